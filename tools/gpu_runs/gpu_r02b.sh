cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 2400 python -m pytest tests -m gpu -q --tb=line -rP 2>&1 | grep -v "^$" | tail -150 > gpurun_out/r02b/pytest.log
cat gpurun_out/r02b/pytest.log | cut -c1-400
