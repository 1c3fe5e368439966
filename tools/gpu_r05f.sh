#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conic_gpu.py tests/test_teacher_forced_gpu.py tests/test_generic_gpu.py tests/test_gusto_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for L in 128 32; do
SCP_CONIC_LONG_ITEM=$L python tools/k5_starship_probe.py 2 30 2>/dev/null | tail -1 | cut -c1-330
SCP_CONIC_LONG_ITEM=$L python tools/k5_starship_probe.py 2 256 2>/dev/null | tail -1 | cut -c1-330
SCP_CONIC_LONG_ITEM=$L python tools/k5_starship_probe.py 2 64 freeflyer 2>/dev/null | tail -1 | cut -c1-360
done
