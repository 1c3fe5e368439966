cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
SCP_CONIC_WAVES=16 timeout 120 python tools/conic_debug.py > gpurun_out/r02j/debug.log 2>&1
SCP_CONIC_WAVES=1 timeout 120 python tools/conic_debug.py >> gpurun_out/r02j/debug.log 2>&1
cat gpurun_out/r02j/debug.log | tail -30
timeout 600 python tools/scvx_debug.py > gpurun_out/r02j/scvx.log 2>&1
tail -60 gpurun_out/r02j/scvx.log
