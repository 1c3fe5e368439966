# MFMA variant of the K3 block products against the vector build: same workload, same iteration counts (warm = 0)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04d}
mkdir -p $OUT
export TMPDIR=/tmp
MF=$GRAFT_REPO_ROOT/scptoolbox.jl_amd/csrc/libscp_mi355x_mfma.so
for rep in 1 2; do
python tools/ipm_iter_stats.py rocket_landing 4096 warm=0 > $OUT/vec_cold_$rep.log 2>&1; head -1 $OUT/vec_cold_$rep.log
SCP_MI355X_LIB=$MF python tools/ipm_iter_stats.py rocket_landing 4096 warm=0 > $OUT/mfma_cold_$rep.log 2>&1; head -1 $OUT/mfma_cold_$rep.log
done
python tools/ipm_iter_stats.py rocket_landing 4096 > $OUT/vec_warm.log 2>&1; head -1 $OUT/vec_warm.log
SCP_MI355X_LIB=$MF python tools/ipm_iter_stats.py rocket_landing 4096 > $OUT/mfma_warm.log 2>&1; head -1 $OUT/mfma_warm.log
tail -17 $OUT/mfma_warm.log
( SCP_MI355X_LIB=$MF timeout 900 python -m pytest tests/test_ptr_gpu.py tests/test_config_size_gpu.py tests/test_failures_gpu.py -q -p no:cacheprovider ) > $OUT/pytest_mfma.log 2>&1
tail -5 $OUT/pytest_mfma.log
