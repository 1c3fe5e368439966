#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp
for S in 3 4 2; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-generic --no-cpu-baseline --streams $S > $O/bench_s$S.json 2> $O/bench_s$S.err
  python -c "
import json;d=json.load(open('$O/bench_s$S.json'));print('streams $S', round(d['value']),round(d['ms_per_step']),d['roofline'].get('avg_launch_ms'), d['roofline'].get('sub_launch_avg_ms'), {k:round(v,2) for k,v in d['kernel_seconds'].items() if k!='note'}, (d.get('to_convergence') or {}).get('value_to_convergence'))"
done
