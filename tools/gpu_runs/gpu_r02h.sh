cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_conic_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/r02h/pytest.log
cat gpurun_out/r02h/pytest.log
for W in 16 8 4; do
SCP_CONIC_WAVES=$W timeout 300 python tools/conic_bench.py conic_rocket_landing_N100 1024 4096 >> gpurun_out/r02h/bench_rocket.json 2>> gpurun_out/r02h/bench_rocket.err
done
cat gpurun_out/r02h/bench_rocket.json; tail -3 gpurun_out/r02h/bench_rocket.err
SCP_CONIC_WAVES=16 timeout 200 python tools/conic_bench.py conic_quadrotor_N50 4096 16384 > gpurun_out/r02h/bench_quad.json 2> gpurun_out/r02h/bench_quad.err
cat gpurun_out/r02h/bench_quad.json; tail -3 gpurun_out/r02h/bench_quad.err
