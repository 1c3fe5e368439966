# usage: bash tools/gpu_r06_pmc.sh <tag>  -- rocprofv3 passes of the headline step: kernel trace, FETCH_SIZE, WRITE_SIZE, SQ counters (separate runs)
TAG=${1:-r06_pmc}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SUM="python $GRAFT_REPO_ROOT/tools/rocpd_summary.py"
HEAD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-solo --no-convergence"
rocprofv3 --kernel-trace --stats -d $OUT/kt -- $HEAD > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
$SUM $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.csv 2>> $OUT/kt.err
head -8 $OUT/kernel_stats.csv
: > $OUT/pmc_hbm.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -- $HEAD > /dev/null 2> $OUT/pmc_$C.err
  $SUM $(find $OUT/pmc_$C -name "*.db" | head -1) | grep -A30 "PMC counters" >> $OUT/pmc_hbm.csv
done
head -4 $OUT/pmc_hbm.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc_sq -- $HEAD > /dev/null 2> $OUT/pmc_sq.err
$SUM $(find $OUT/pmc_sq -name "*.db" | head -1) | grep -A40 "PMC counters" > $OUT/sq_counters.csv
head -10 $OUT/sq_counters.csv
if [ -n "$EXTRA_PMC" ]; then
rocprofv3 --kernel-trace --pmc $EXTRA_PMC -d $OUT/pmc_x -- $HEAD > /dev/null 2> $OUT/pmc_x.err
$SUM $(find $OUT/pmc_x -name "*.db" | head -1) | grep -A40 "PMC counters" > $OUT/x_counters.csv
head -12 $OUT/x_counters.csv
fi
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq $OUT/pmc_x
