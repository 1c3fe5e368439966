#!/bin/bash
# round 5 final: default bench + rocprofv3 evidence (kernel trace, HBM PMC passes, SQ counters) for K3 / K1v (headline), K1 (reference form), K5 (one workgroup per problem)
TAG=${1:-r05_final}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$SKIP_BENCH" ]; then
( time python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
head -c 400 $OUT/bench.json; echo; tail -3 $OUT/bench.err
fi
if [ -n "$SKIP_PROF" ]; then exit 0; fi
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
if [ -n "$ONLY_HEAD" ]; then rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq; exit 0; fi
# ---- K5, one workgroup per problem: the config-3 program (Starship SCvx N = 100), 256 problems ----
CB="python $GRAFT_REPO_ROOT/tools/k5_starship_probe.py 1 256"
rocprofv3 --kernel-trace --stats -d $OUT/k5kt -- $CB > $OUT/k5_probe.json 2> $OUT/k5kt.err
$SUM $(find $OUT/k5kt -name "*.db" | head -1) | head -6 > $OUT/k5_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/k5sq -- $CB > /dev/null 2> $OUT/k5sq.err
$SUM $(find $OUT/k5sq -name "*.db" | head -1) | grep -A12 "PMC counters" > $OUT/k5_sq_counters.csv
head -5 $OUT/k5_kernel_stats.csv; head -6 $OUT/k5_sq_counters.csv
SCP_MI355X_LIB=$GRAFT_REPO_ROOT/scptoolbox.jl_amd/csrc/libscp_mi355x_cprof.so python $GRAFT_REPO_ROOT/tools/k5_starship_probe.py 1 30 2>&1 | grep -a "CONIC_PROF" > $OUT/k5_phase_profile.txt
cat $OUT/k5_phase_profile.txt
# ---- K1 reference form (free-flyer N = 200 x 4096, Starship N = 100 x 256) ----
K1="python $GRAFT_REPO_ROOT/tools/k1_bench.py"
rocprofv3 --kernel-trace --stats -d $OUT/k1kt -- $K1 > $OUT/k1_bench.json 2> $OUT/k1kt.err
$SUM $(find $OUT/k1kt -name "*.db" | head -1) | head -8 > $OUT/k1_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/k1sq -- $K1 > /dev/null 2> $OUT/k1sq.err
$SUM $(find $OUT/k1sq -name "*.db" | head -1) | grep -A14 "PMC counters" > $OUT/k1_sq_counters.csv
head -8 $OUT/k1_kernel_stats.csv; head -8 $OUT/k1_sq_counters.csv
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq $OUT/k5kt $OUT/k5sq $OUT/k1kt $OUT/k1sq
