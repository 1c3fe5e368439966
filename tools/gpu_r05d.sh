#!/bin/bash
# round 5, call 4: stream priorities A/B on the headline, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
for P in 0 1; do
  SCP_STREAM_PRIORITIES=$P timeout 300 python bench.py --steps 3 --warmup 1 --no-generic --no-cpu-baseline > $O/bench_prio$P.json 2> $O/bench_prio$P.err
  python -c "
import json;d=json.load(open('$O/bench_prio$P.json'));print('prio $P', d['value'],d['ms_per_step'],d['roofline'].get('avg_launch_ms'), d['roofline'].get('sub_launch_avg_ms'), d['kernel_seconds'], d.get('to_convergence'))"
done
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 ) > $O/pytest.log 2>&1
tail -22 $O/pytest.log
