cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
export SCP_IPM_WPE=2
for cfg in "8 15" "4 15" "2 15" "8 1" "32 15"; do
  set -- $cfg
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --streams $1 --lookahead $2 > gpurun_out/r02f/bench_$1_$2.json 2> gpurun_out/r02f/bench_$1_$2.err
  python -c "
import json;d=json.load(open('gpurun_out/r02f/bench_$1_$2.json'));print('WPE2 streams $1 lookahead $2:', round(d['value']), round(d['ms_per_step']), d['roofline']['avg_launch_ms'], d['roofline']['sub_launch_avg_ms'], d['roofline']['ipm_iterations_mean'], d['failed_instances'])"
done
