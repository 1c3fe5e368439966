# usage: bash tools/gpu_final.sh <tag>   -- full GPU test suite, bench, rocprofv3 kernel trace + PMC passes (round 3)
TAG=${1:-r03_final}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
( time python bench.py --steps 3 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err
head -c 600 $OUT/bench.json; echo; tail -4 $OUT/bench.err
cd /tmp
# the SAME command under the profiler (headline workload only: the sub-records have their own profiles)
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-solo > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.csv 2>> $OUT/kt.err
head -14 $OUT/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-solo > /dev/null 2> $OUT/pmc_$C.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $OUT/pmc_$C -name "*.db" | head -1) | grep -A30 "PMC counters" > $OUT/pmc_$C.csv
  head -6 $OUT/pmc_$C.csv
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-solo > /dev/null 2> $OUT/pmc_sq.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $OUT/pmc_sq -name "*.db" | head -1) | grep -A40 "PMC counters" > $OUT/pmc_sq.csv
head -8 $OUT/pmc_sq.csv
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq
cd $GRAFT_REPO_ROOT
# BASELINE.json configs[2] to the end of the reference's stopping rule (bounded: 420 s)
[ -n "$SKIP_STARSHIP" ] || timeout 600 python tools/starship_n100.py 256 $OUT/starship_n100_scvx.json 420 > $OUT/starship_n100.log 2>&1
[ -n "$SKIP_STARSHIP" ] || tail -c 600 $OUT/starship_n100.log
