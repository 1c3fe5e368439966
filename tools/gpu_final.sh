# usage: bash tools/gpu_final.sh <tag>   -- full GPU test suite, bench, rocprofv3 kernel trace + PMC passes, Starship N=100
TAG=${1:-r02_final}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; tail -3 $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.csv 2>> $OUT/kt.err
head -20 $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic > /dev/null 2> $OUT/pmc_$C.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $OUT/pmc_$C -name "*.db" | head -1) | grep -A30 "PMC counters" > $OUT/pmc_$C.csv
  head -4 $OUT/pmc_$C.csv
done
cd $GRAFT_REPO_ROOT
timeout 420 python tools/starship_n100.py 8 16 $OUT/starship_n100_scvx.json > $OUT/starship_n100.log 2>&1
tail -2 $OUT/starship_n100.log | cut -c1-1500
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/cpmc_$C -- python $GRAFT_REPO_ROOT/tools/conic_bench.py conic_rocket_landing_N100 16384 > $OUT/conic_bench_$C.json 2> $OUT/cpmc_$C.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $OUT/cpmc_$C -name "*.db" | head -1) | grep -A10 "PMC counters" > $OUT/conic_pmc_$C.csv
  head -4 $OUT/conic_pmc_$C.csv
done
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/cpmc_FETCH_SIZE $OUT/cpmc_WRITE_SIZE
