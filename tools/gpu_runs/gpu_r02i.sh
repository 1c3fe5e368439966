cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
for W in 1 2 16; do SCP_CONIC_WAVES=$W timeout 120 python tools/conic_debug.py >> gpurun_out/r02i/debug.log 2>&1; done
cat gpurun_out/r02i/debug.log | tail -40
timeout 900 python -m pytest tests/test_conic_gpu.py tests/test_generic_gpu.py -q 2>&1 | tail -40 > gpurun_out/r02i/pytest.log
cat gpurun_out/r02i/pytest.log
