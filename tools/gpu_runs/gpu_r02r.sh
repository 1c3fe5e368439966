cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02r
timeout 900 python -m pytest tests/test_conic_gpu.py tests/test_generic_gpu.py tests/test_gusto_gpu.py -q -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/r02r/pytest.log
cat gpurun_out/r02r/pytest.log
timeout 300 python tools/conic_bench.py conic_rocket_landing_N100 1024 4096 16384 > gpurun_out/r02r/bench_rocket.json 2> gpurun_out/r02r/bench_rocket.err
SCP_CONIC_ORDER=seq timeout 300 python tools/conic_bench.py conic_rocket_landing_N100 1024 >> gpurun_out/r02r/bench_rocket.json 2>> gpurun_out/r02r/bench_rocket.err
cut -c1-420 gpurun_out/r02r/bench_rocket.json; tail -3 gpurun_out/r02r/bench_rocket.err
