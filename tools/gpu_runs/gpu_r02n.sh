cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02n
timeout 600 python tools/starship_debug.py > gpurun_out/r02n/debug.log 2>&1
tail -70 gpurun_out/r02n/debug.log
