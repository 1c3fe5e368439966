#!/bin/bash
# round 5: K5 at the full-chip geometry (16 384 literal rocket programs) after the regularisation change: kernel trace, HBM counters, SQ counters
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_k5; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SUM="python $GRAFT_REPO_ROOT/tools/rocpd_summary.py"
B=16384
CB="python $GRAFT_REPO_ROOT/tools/conic_bench.py conic_rocket_landing_N100 $B"
rocprofv3 --kernel-trace --stats -d $OUT/k5kt -- $CB > $OUT/conic_bench_$B.json 2> $OUT/k5kt.err
$SUM $(find $OUT/k5kt -name "*.db" | head -1) | head -6 > $OUT/conic_kernel_stats_$B.csv
: > $OUT/conic_pmc_hbm_$B.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/k5pmc_$C -- $CB > /dev/null 2> $OUT/k5pmc_$C.err
  $SUM $(find $OUT/k5pmc_$C -name "*.db" | head -1) | grep -A6 "PMC counters" >> $OUT/conic_pmc_hbm_$B.csv
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/k5sq -- $CB > /dev/null 2> $OUT/k5sq.err
$SUM $(find $OUT/k5sq -name "*.db" | head -1) | grep -A12 "PMC counters" > $OUT/conic_sq_counters_$B.csv
cat $OUT/conic_bench_$B.json | head -c 500; echo; cat $OUT/conic_kernel_stats_$B.csv; cat $OUT/conic_pmc_hbm_$B.csv; cat $OUT/conic_sq_counters_$B.csv
rm -rf $OUT/k5kt $OUT/k5pmc_FETCH_SIZE $OUT/k5pmc_WRITE_SIZE $OUT/k5sq
