# first GPU call of round 4: parity of the new warm start / regularisation + per-iteration K3 statistics + a short bench
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04a}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_outcomes_gpu.py tests/test_ptr_gpu.py tests/test_config_size_gpu.py tests/test_failures_gpu.py tests/test_shim_sequence_gpu.py tests/test_generic_gpu.py tests/test_gusto_gpu.py -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -30 $OUT/pytest.log
python tools/ipm_iter_stats.py rocket_landing 4096 > $OUT/iter_stats.log 2>&1
cat $OUT/iter_stats.log | tail -20
python tools/ipm_iter_stats.py rocket_landing 4096 warm=0,reg=5e-11 > $OUT/iter_stats_cold_oldreg.log 2>&1
tail -18 $OUT/iter_stats_cold_oldreg.log
( time python bench.py --steps 2 --warmup 1 --no-generic --no-cpu-baseline ) > $OUT/bench.json 2> $OUT/bench.err
head -c 1500 $OUT/bench.json; echo; tail -3 $OUT/bench.err
