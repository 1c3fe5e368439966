#!/bin/bash
# usage: bash tools/gpu_k3_timeline.sh <tag> -- where the headline step's time goes BETWEEN the K3 waves: (1) the IPM iteration count of every
# problem in every launch + the launch times (input of tools/k3_packing_sim.py), (2) the dispatch timeline of one bench step (which launches
# overlap, where a stream waits for its stragglers)
TAG=${1:-r06_timeline}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
K3_ITERS_DUMP=$OUT/k3_iters_rocket_landing_4096.npz python tools/ipm_iter_stats.py rocket_landing 4096 > $OUT/iter_stats.txt 2>&1
cat $OUT/iter_stats.txt
cd /tmp
rocprofv3 --kernel-trace -d $OUT/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-solo --no-convergence > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py --timeline $(find $OUT/kt -name "*.db" | head -1) | grep -v "rocclr" > $OUT/timeline.csv
wc -l $OUT/timeline.csv; grep ipm2 $OUT/timeline.csv | head -40
rm -rf $OUT/kt
