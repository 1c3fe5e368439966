cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02m
timeout 1500 python -m pytest tests/test_starship_gpu.py tests/test_generic_gpu.py tests/test_conic_gpu.py -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r02m/pytest.log
cat gpurun_out/r02m/pytest.log
