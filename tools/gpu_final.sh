#!/bin/bash
# usage: bash tools/gpu_final.sh <tag> -- the round's evidence on ONE box (run through gpurun): GPU test suite, the bench line with the DRIVER's
# command, rocprofv3 passes (kernel trace / FETCH_SIZE / WRITE_SIZE / SQ counters, separate runs) for every kernel a record prices:
# K3 + K1v (headline step), K5 (full chip, one workgroup per problem, the free-flyer geometry), K1 (reference form), K3 phase clocks, and
# BASELINE.json configs[2] run to iter_max.  Environment: SKIP_TESTS / SKIP_BENCH / SKIP_PROF / ONLY_HEAD = 1 to leave parts out.
TAG=${1:-r06_final}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
( time timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
fi
if [ -z "$SKIP_BENCH" ]; then
( time python bench.py ${BENCH_ARGS:---steps 20 --warmup 5} ) > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json; echo; tail -3 $OUT/bench.err
cp bench_records.json $OUT/bench_records.json 2>/dev/null
fi
if [ -n "$SKIP_PROF" ]; then exit 0; fi
cd /tmp
SUM="python $GRAFT_REPO_ROOT/tools/rocpd_summary.py"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
passes() {   # passes <prefix> <grep -A lines> -- <command>: kernel trace + FETCH / WRITE / SQ counter passes of one command
  local pre=$1 nl=$2; shift 3
  rocprofv3 --kernel-trace --stats -d $OUT/${pre}kt -- "$@" > $OUT/${pre}bench_under_rocprof.json 2> $OUT/${pre}kt.err
  $SUM $(find $OUT/${pre}kt -name "*.db" | head -1) | head -12 > $OUT/${pre}kernel_stats.csv
  : > $OUT/${pre}pmc_hbm.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C -d $OUT/${pre}pmc_$C -- "$@" > /dev/null 2> $OUT/${pre}pmc_$C.err
    $SUM $(find $OUT/${pre}pmc_$C -name "*.db" | head -1) | grep -A$nl "PMC counters" >> $OUT/${pre}pmc_hbm.csv
  done
  rocprofv3 --kernel-trace --pmc $SQ -d $OUT/${pre}pmc_sq -- "$@" > /dev/null 2> $OUT/${pre}pmc_sq.err
  $SUM $(find $OUT/${pre}pmc_sq -name "*.db" | head -1) | grep -A40 "PMC counters" > $OUT/${pre}sq_counters.csv
  head -5 $OUT/${pre}kernel_stats.csv; head -4 $OUT/${pre}pmc_hbm.csv; head -9 $OUT/${pre}sq_counters.csv
  rm -rf $OUT/${pre}kt $OUT/${pre}pmc_FETCH_SIZE $OUT/${pre}pmc_WRITE_SIZE $OUT/${pre}pmc_sq
}
# ---- headline workload (K3, K1v, K2, K4) ----
passes "" 30 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-solo --no-convergence
# ---- K3 phase clocks (make prof) ----
if [ -f $GRAFT_REPO_ROOT/scptoolbox.jl_amd/csrc/libscp_mi355x_prof.so ]; then
  ( cd $GRAFT_REPO_ROOT; for IT in 1 10; do SCP_MI355X_LIB=$GRAFT_REPO_ROOT/scptoolbox.jl_amd/csrc/libscp_mi355x_prof.so python tools/ipm_phase_profile.py rocket_landing 4096 $IT; done ) > $OUT/k3_phase_profile.txt 2>&1
  cat $OUT/k3_phase_profile.txt
fi
if [ -n "$ONLY_HEAD" ]; then exit 0; fi
# ---- K5: full chip (16 384 rocket programs), one workgroup per problem (Starship SCvx N = 100 x 256), the free-flyer GuSTO geometry (N = 200 x 512) ----
passes "conic_16384_" 6 -- python $GRAFT_REPO_ROOT/tools/conic_bench.py conic_rocket_landing_N100 16384
passes "k5_starship_" 6 -- python $GRAFT_REPO_ROOT/tools/k5_starship_probe.py 1 256
passes "k5_freeflyer_" 6 -- python $GRAFT_REPO_ROOT/tools/k5_starship_probe.py 1 512 freeflyer
# ---- K1 reference form (free-flyer N = 200 x 4096, Starship N = 100 x 256) ----
passes "k1_" 14 -- python $GRAFT_REPO_ROOT/tools/k1_bench.py
# ---- BASELINE.json configs[2] to the reference's iter_max ----
( cd $GRAFT_REPO_ROOT; python tools/starship_n100.py 256 $OUT/starship_n100_scvx_256_100iters.json 600 > /dev/null 2> $OUT/starship_n100.err; python -c "
import json; d=json.load(open('$OUT/starship_n100_scvx_256_100iters.json')); print({k: d[k] for k in ('loop_iterations','seconds_per_loop_iteration','frac_failed','frac_converged','frac_dyn_feasible','guess_seconds')}); print(d.get('oracle_monte_carlo'))" )
