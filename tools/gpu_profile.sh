# usage: bash tools/gpu_profile.sh <tag> [pmc]   -- kernel trace + stats of one bench step (and optional PMC passes)
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.csv 2>> $OUT/kt.err
head -12 $OUT/kernel_stats.csv
if [ "$2" = "pmc" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$C.err
    python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $OUT/pmc_$C -name "*.db" | head -1) | grep -A30 "PMC counters" > $OUT/pmc_$C.csv
    head -5 $OUT/pmc_$C.csv
  done
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $OUT/pmc_sq -name "*.db" | head -1) | grep -A40 "PMC counters" > $OUT/pmc_sq.csv
  head -12 $OUT/pmc_sq.csv
fi
rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq
