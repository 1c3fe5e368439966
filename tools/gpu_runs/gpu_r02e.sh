cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
timeout 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15 > gpurun_out/r02e/pytest.log
cat gpurun_out/r02e/pytest.log
for cfg in "1 1" "8 1" "8 15" "4 15" "16 15"; do
  set -- $cfg
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --streams $1 --lookahead $2 > gpurun_out/r02e/bench_$1_$2.json 2> gpurun_out/r02e/bench_$1_$2.err
  python -c "
import json;d=json.load(open('gpurun_out/r02e/bench_$1_$2.json'));print('streams $1 lookahead $2:', round(d['value']), round(d['ms_per_step']), d['roofline']['avg_launch_ms'], d['roofline']['sub_launch_avg_ms'], d['roofline']['ipm_iterations_mean'], d['failed_instances'])"
done
