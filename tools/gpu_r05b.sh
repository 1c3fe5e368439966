#!/bin/bash
# round 5, call 2: Starship guess parity + 30-iteration loops from both oracle records + K1 (structural LU lead, stage loop): parity and timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_starship_gpu.py tests/test_discretize_gpu.py tests/test_freeflyer_gpu.py "tests/test_teacher_forced_gpu.py::test_starship_scvx_subproblems_at_config_size_about_the_oracles_references" \
   tests/test_generic_gpu.py -m gpu -q --durations=8 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
cp gpurun_out/*.json $O/ 2>/dev/null
timeout 300 python tools/k1_bench.py > $O/k1_bench.json 2> $O/k1_bench.err; echo "k1 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05b/k1_bench.json"))
f=d["freeflyer"]; s=d["starship"]
print({k:f[k] for k in f if "ms" in k or "seconds" in k or "frac" in k})
print({k:s[k] for k in s if "ms" in k or "seconds" in k})
PY
