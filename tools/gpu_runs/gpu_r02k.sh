cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
timeout 120 python tools/conic_debug2.py > gpurun_out/r02k/debug.log 2>&1
cat gpurun_out/r02k/debug.log | tail -14
