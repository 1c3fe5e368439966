# configs 3 and 5 at their stated sizes (long runs), stream sweep of the headline bench
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04e}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python tools/freeflyer_n200.py 512 $OUT/freeflyer_n200_b512.json 15 ) > $OUT/freeflyer_n200.log 2>&1
tail -c 1500 $OUT/freeflyer_n200.log
( time timeout 700 python tools/starship_n100.py 256 $OUT/starship_n100_scvx.json 420 ) > $OUT/starship_n100.log 2>&1
tail -c 1800 $OUT/starship_n100.log
for S in 3 4; do
python bench.py --steps 2 --warmup 1 --no-generic --no-cpu-baseline --streams $S > $OUT/bench_streams$S.json 2>/dev/null; head -c 330 $OUT/bench_streams$S.json; echo
done
