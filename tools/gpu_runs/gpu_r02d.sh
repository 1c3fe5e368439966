cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 2400 python -m pytest tests -m gpu -q --tb=line -rP 2>&1 | grep -v "^$" | grep -E "device vs|passed|failed|Error|error|assert" | cut -c1-300 > gpurun_out/r02d/pytest.log
cat gpurun_out/r02d/pytest.log
python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err
python -c "
import json;d=json.load(open('gpurun_out/r02d/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['ipm_iterations_mean'], d['failed_instances'], d['residual'])"
tail -3 gpurun_out/r02d/bench.err
