# full GPU suite + default bench
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04c}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log
( time python bench.py --steps 3 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err
head -c 2500 $OUT/bench.json; echo; tail -3 $OUT/bench.err
