# usage: bash tools/gpu_r06_base.sh <tag>  -- GPU test suite, default bench line, K3 phase profile (prof build)
TAG=${1:-r06_base}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
fi
if [ -z "$SKIP_BENCH" ]; then
( time python bench.py ${BENCH_ARGS:---steps 3 --warmup 1} ) > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json; echo; tail -4 $OUT/bench.err
cp bench_records.json $OUT/bench_records.json 2>/dev/null
fi
if [ -f scptoolbox.jl_amd/csrc/libscp_mi355x_prof.so ]; then
for IT in 1 10; do
SCP_MI355X_LIB=$GRAFT_REPO_ROOT/scptoolbox.jl_amd/csrc/libscp_mi355x_prof.so python tools/ipm_phase_profile.py rocket_landing 4096 $IT > $OUT/k3_phase_$IT.txt 2>&1
cat $OUT/k3_phase_$IT.txt
done
fi
