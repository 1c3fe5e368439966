cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02q
timeout 900 python -m pytest tests/test_gusto_gpu.py tests/test_starship_gpu.py::test_fp32_discretize_tolerance_check tests/test_starship_gpu.py::test_fp32_discretize_is_refused_where_it_does_not_exist tests/test_starship_gpu.py::test_discretize_matches_oracle_at_the_reference_config -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r02q/pytest.log
cat gpurun_out/r02q/pytest.log
timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r02q/bench.json 2> gpurun_out/r02q/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r02q/bench.json"))
print(r["value"], r["ms_per_step"])
print(json.dumps(r.get("generic_path", {}), indent=1)[:5000])
PY
tail -5 gpurun_out/r02q/bench.err
