#!/bin/bash
# round 5: whole GPU suite + default bench on the tree with the fused K4a, auto_reg 1e-10 / dyn_delta 2e-6 and the new teacher-forced tests
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05i}; mkdir -p $OUT
( time timeout 900 python -m pytest tests -q -m gpu ${PYTEST_ARGS:--x} ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
if [ -z "$SKIP_BENCH" ]; then
( time python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
head -c 300 $OUT/bench.json; echo; tail -3 $OUT/bench.err
fi
