cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02o
timeout 600 python -m pytest tests/test_conic_gpu.py "tests/test_generic_gpu.py::test_compute_scaling_on_device_matches_the_analytic_boxes" "tests/test_starship_gpu.py::test_reference_guess_on_device_matches_golden" -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r02o/pytest.log
cat gpurun_out/r02o/pytest.log
for S in 1 4 16; do
SCP_CONIC_SUB=$S timeout 300 python tools/conic_bench.py conic_rocket_landing_N100 1024 4096 >> gpurun_out/r02o/bench_rocket.json 2>> gpurun_out/r02o/bench_rocket.err
done
SCP_CONIC_SUB=4 SCP_CONIC_WAVES=8 timeout 300 python tools/conic_bench.py conic_rocket_landing_N100 4096 >> gpurun_out/r02o/bench_rocket.json 2>> gpurun_out/r02o/bench_rocket.err
cat gpurun_out/r02o/bench_rocket.json | cut -c1-330; tail -3 gpurun_out/r02o/bench_rocket.err
