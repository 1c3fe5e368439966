set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
nproc > gpurun_out/r02a/nproc.txt; lscpu | head -20 >> gpurun_out/r02a/nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r02a/pytest.log
cat gpurun_out/r02a/pytest.log
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
cat gpurun_out/r02a/bench.json
