cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02p
timeout 900 python -m pytest tests/test_gusto_gpu.py "tests/test_generic_gpu.py::test_compute_scaling_on_device_matches_the_analytic_boxes" tests/test_starship_gpu.py tests/test_abi.py -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r02p/pytest.log
cat gpurun_out/r02p/pytest.log
