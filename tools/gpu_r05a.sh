#!/bin/bash
# round 5, call 1: the new parity tests (teacher-forced subproblems, 30-iteration Starship loop, C-ABI sharded loop) + a quick headline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_teacher_forced_gpu.py "tests/test_dist_gpu.py::test_sharded_loop_behind_the_c_abi_with_a_one_rank_rccl_communicator" \
    "tests/test_starship_gpu.py::test_scvx_thirty_iterations_at_config_size_follow_the_oracle" -m gpu -q -x --durations=10 > gpurun_out/r05a/pytest_new.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05a/pytest_new.log
tail -30 gpurun_out/r05a/pytest_new.log
cp gpurun_out/teacher_forced_*.json gpurun_out/starship_scvx_N100_30_iterations.json gpurun_out/r05a/ 2>/dev/null
timeout 300 python bench.py --steps 3 --warmup 1 --no-generic --no-cpu-baseline > gpurun_out/r05a/bench_quick.json 2> gpurun_out/r05a/bench_quick.err
echo "bench rc=$?"; python -c "
import json;d=json.load(open('gpurun_out/r05a/bench_quick.json'));print(d['value'],d['ms_per_step'],d['roofline'].get('avg_launch_ms'))"
