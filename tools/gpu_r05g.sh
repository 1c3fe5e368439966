#!/bin/bash
# round 5: K4a fused into the tail of the solving wave -- A/B on one box (SCP_K3_FUSE_EXTRACT=0 is the separate launch), 2 and 3 streams, PTR tests
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05g; mkdir -p $OUT
H="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-generic --no-solo --no-convergence"
for F in 0 1 0 1; do
  SCP_K3_FUSE_EXTRACT=$F $H > $OUT/bench_fuse$F.json 2> $OUT/bench_fuse$F.err
  python -c "import json,sys; d=json.load(open('$OUT/bench_fuse$F.json')); print('fuse', $F, d['value'], d['ms_per_step'], d['roofline']['sub_launch_avg_ms'])"
done
SCP_K3_FUSE_EXTRACT=1 $H --streams 3 > $OUT/bench_fuse1_s3.json 2> $OUT/bench_fuse1_s3.err
python -c "import json,sys; d=json.load(open('$OUT/bench_fuse1_s3.json')); print('fuse 1 streams 3', d['value'], d['ms_per_step'], d['roofline']['sub_launch_avg_ms'])"
SCP_K3_FUSE_EXTRACT=1 $H --streams 4 > $OUT/bench_fuse1_s4.json 2> $OUT/bench_fuse1_s4.err
python -c "import json,sys; d=json.load(open('$OUT/bench_fuse1_s4.json')); print('fuse 1 streams 4', d['value'], d['ms_per_step'], d['roofline']['sub_launch_avg_ms'])"
( time timeout 600 python -m pytest tests/test_ptr_gpu.py tests/test_golden_gpu.py tests/test_subproblem_gpu.py tests/test_failures_gpu.py -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
