cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02s
timeout 300 python -m pytest tests/test_gusto_gpu.py -q -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r02s/pytest.log
cat gpurun_out/r02s/pytest.log
for S in 1 4 16; do
SCP_CONIC_SUB=$S timeout 200 python tools/conic_bench.py conic_rocket_landing_N100 256 1024 4096 >> gpurun_out/r02s/bench_sub.json 2>> gpurun_out/r02s/bench.err
done
SCP_CONIC_SUB=1 SCP_CONIC_WAVES=8 timeout 200 python tools/conic_bench.py conic_rocket_landing_N100 4096 16384 >> gpurun_out/r02s/bench_sub.json 2>> gpurun_out/r02s/bench.err
python - <<'PY'
import json
for l in open("gpurun_out/r02s/bench_sub.json"):
    r = json.loads(l); print(r["B"], "waves", r["stats"]["waves"], "seconds %.3f" % r["seconds"], "pps %.0f" % r["problems_per_s"])
PY
tail -3 gpurun_out/r02s/bench.err
