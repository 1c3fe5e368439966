# usage: bash tools/gpu_final.sh <tag>   -- bench + rocprofv3 kernel trace + PMC passes for every kernel a record prices (round 4)
TAG=${1:-r04_final}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
fi
( time python bench.py --steps 3 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err
head -c 600 $OUT/bench.json; echo; tail -4 $OUT/bench.err
cd /tmp
SUM="python $GRAFT_REPO_ROOT/tools/rocpd_summary.py"
HEAD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-solo"
prof() {   # prof <name> <rocprof args...> -- <command>
  local name=$1; shift
  rocprofv3 "$@" > $OUT/$name.out 2> $OUT/$name.err
}
# ---- headline workload (K3, K1v, K2, K4) ----
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
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc_mfma -- $HEAD > /dev/null 2> $OUT/pmc_mfma.err
$SUM $(find $OUT/pmc_mfma -name "*.db" | head -1) | grep -A12 "PMC counters" > $OUT/mfma_counters.csv
head -6 $OUT/mfma_counters.csv
# ---- K5 at both geometries: 16 384 problems interleaved (SUB = 1) and 256 problems, one per workgroup (SUB = 64) ----
for B in 16384 256; do
  CB="python $GRAFT_REPO_ROOT/tools/conic_bench.py conic_rocket_landing_N100 $B"
  rocprofv3 --kernel-trace --stats -d $OUT/k5kt_$B -- $CB > $OUT/conic_bench_$B.json 2> $OUT/k5kt_$B.err
  $SUM $(find $OUT/k5kt_$B -name "*.db" | head -1) | head -6 > $OUT/conic_kernel_stats_$B.csv
  : > $OUT/conic_pmc_hbm_$B.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C -d $OUT/k5pmc_${C}_$B -- $CB > /dev/null 2> $OUT/k5pmc_${C}_$B.err
    $SUM $(find $OUT/k5pmc_${C}_$B -name "*.db" | head -1) | grep -A6 "PMC counters" >> $OUT/conic_pmc_hbm_$B.csv
  done
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/k5sq_$B -- $CB > /dev/null 2> $OUT/k5sq_$B.err
  $SUM $(find $OUT/k5sq_$B -name "*.db" | head -1) | grep -A12 "PMC counters" > $OUT/conic_sq_counters_$B.csv
  cat $OUT/conic_bench_$B.json | head -c 400; echo
  head -4 $OUT/conic_pmc_hbm_$B.csv
done
# ---- K1 / K1x on state-dependent Jacobians (freeflyer N = 200 x 4096, Starship N = 100 x 256) ----
K1="python $GRAFT_REPO_ROOT/tools/k1_bench.py"
rocprofv3 --kernel-trace --stats -d $OUT/k1kt -- $K1 > $OUT/k1_bench.json 2> $OUT/k1kt.err
$SUM $(find $OUT/k1kt -name "*.db" | head -1) | head -12 > $OUT/k1_kernel_stats.csv
head -8 $OUT/k1_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/k1sq -- $K1 > /dev/null 2> $OUT/k1sq.err
$SUM $(find $OUT/k1sq -name "*.db" | head -1) | grep -A30 "PMC counters" > $OUT/k1_sq_counters.csv
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq $OUT/pmc_mfma $OUT/k5kt_* $OUT/k5pmc_* $OUT/k5sq_* $OUT/k1kt $OUT/k1sq
cd $GRAFT_REPO_ROOT
