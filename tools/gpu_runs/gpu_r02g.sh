cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
timeout 600 python -m pytest tests/test_conic_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r02g/pytest.log
cat gpurun_out/r02g/pytest.log
timeout 400 python tools/conic_bench.py conic_rocket_landing_N100 1024 4096 16384 > gpurun_out/r02g/bench_rocket.json 2> gpurun_out/r02g/bench_rocket.err
cat gpurun_out/r02g/bench_rocket.json; tail -3 gpurun_out/r02g/bench_rocket.err
timeout 200 python tools/conic_bench.py conic_quadrotor_N50 4096 16384 > gpurun_out/r02g/bench_quad.json 2> gpurun_out/r02g/bench_quad.err
cat gpurun_out/r02g/bench_quad.json; tail -3 gpurun_out/r02g/bench_quad.err
