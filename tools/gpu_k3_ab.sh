# usage: bash tools/gpu_r06_ab.sh <tag>  -- K3 A/B on ONE box: phase profiles of the baseline prof build (libscp_mi355x_baseprof.so, built
# from an older commit by hand) and of the current one, then the GPU tests and a short bench
TAG=${1:-r06_ab}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for V in ${VARIANTS:-baseprof prof profc}; do
  L=$GRAFT_REPO_ROOT/scptoolbox.jl_amd/csrc/libscp_mi355x_$V.so
  [ -f $L ] || continue
  for IT in 1 10; do
    SCP_MI355X_LIB=$L python tools/ipm_phase_profile.py rocket_landing 4096 $IT > $OUT/k3_phase_${V}_$IT.txt 2>&1
    echo "== $V $IT"; cat $OUT/k3_phase_${V}_$IT.txt
  done
done
if [ -z "$SKIP_TESTS" ]; then
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x ${TEST_ARGS} ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
fi
if [ -z "$SKIP_BENCH" ]; then
( time python bench.py ${BENCH_ARGS:---steps 2 --warmup 1 --no-generic --no-cpu-baseline} ) > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json; echo; tail -4 $OUT/bench.err
fi
