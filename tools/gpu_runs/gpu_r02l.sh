cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02l
timeout 1200 python -m pytest tests/test_conic_gpu.py tests/test_generic_gpu.py tests/test_errors_gpu.py -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r02l/pytest.log
cat gpurun_out/r02l/pytest.log
