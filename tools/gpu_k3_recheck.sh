# usage: bash tools/gpu_k3_recheck.sh <tag>  -- after a change of the K3 sources: its tests, the per-iteration statistics, the bench line
# and the K3 counters again (bench.py reports roofline.traffic only for the source hash the counters were taken on)
TAG=${1:-r04_k3}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_ptr_gpu.py tests/test_config_size_gpu.py tests/test_failures_gpu.py tests/test_outcomes_gpu.py tests/test_dist_gpu.py tests/test_shim_sequence_gpu.py ${EXTRA_TESTS} -m gpu -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
python tools/ipm_iter_stats.py rocket_landing 4096 > $OUT/iter_stats.log 2>&1
head -1 $OUT/iter_stats.log
( time python bench.py --steps 3 --warmup 1 ) > $OUT/bench.json 2> $OUT/bench.err
head -c 300 $OUT/bench.json; echo
cd /tmp
SUM="python $GRAFT_REPO_ROOT/tools/rocpd_summary.py"
HEAD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-generic --no-solo"
rocprofv3 --kernel-trace --stats -d $OUT/kt -- $HEAD > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
$SUM $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.csv 2>> $OUT/kt.err
head -6 $OUT/kernel_stats.csv
: > $OUT/pmc_hbm.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -- $HEAD > /dev/null 2> $OUT/pmc_$C.err
  $SUM $(find $OUT/pmc_$C -name "*.db" | head -1) | grep -A30 "PMC counters" >> $OUT/pmc_hbm.csv
done
head -4 $OUT/pmc_hbm.csv
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc_sq -- $HEAD > /dev/null 2> $OUT/pmc_sq.err
$SUM $(find $OUT/pmc_sq -name "*.db" | head -1) | grep -A40 "PMC counters" > $OUT/sq_counters.csv
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq
cd $GRAFT_REPO_ROOT
