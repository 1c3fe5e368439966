cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02u
mkdir -p $OUT
timeout 170 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/sq -- python $GRAFT_REPO_ROOT/tools/conic_bench.py conic_rocket_landing_N100 16384 > $OUT/conic_bench_sq.json 2> $OUT/sq.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $OUT/sq -name "*.db" | head -1) | grep -A14 "PMC counters" > $OUT/conic_sq_counters.csv
cat $OUT/conic_sq_counters.csv | cut -c1-200
tail -2 $OUT/sq.err
rm -rf $OUT/sq
