# usage: bash tools/gpu_last.sh <tag>  -- last run of a round on the final tree: full GPU suite, the default bench line, and the K1
# kernel trace + SQ counters (the K3 / K5 counters are tied to their sources' hashes and are re-taken only when those change:
# tools/gpu_final.sh, tools/gpu_k3_recheck.sh)
TAG=${1:-r04_last}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
( time python bench.py --steps 3 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err
head -c 300 $OUT/bench.json; echo; tail -4 $OUT/bench.err
cd /tmp
SUM="python $GRAFT_REPO_ROOT/tools/rocpd_summary.py"
K1="python $GRAFT_REPO_ROOT/tools/k1_bench.py"
rocprofv3 --kernel-trace --stats -d $OUT/k1kt -- $K1 > $OUT/k1_bench.json 2> $OUT/k1kt.err
$SUM $(find $OUT/k1kt -name "*.db" | head -1) | head -12 > $OUT/k1_kernel_stats.csv
head -6 $OUT/k1_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/k1sq -- $K1 > /dev/null 2> $OUT/k1sq.err
$SUM $(find $OUT/k1sq -name "*.db" | head -1) | grep -A30 "PMC counters" > $OUT/k1_sq_counters.csv
rm -rf $OUT/k1kt $OUT/k1sq
cd $GRAFT_REPO_ROOT
