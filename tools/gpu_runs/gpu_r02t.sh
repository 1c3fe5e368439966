cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02t
mkdir -p $OUT
timeout 900 python -m pytest tests/test_freeflyer_gpu.py tests/test_starship_gpu.py tests/test_gusto_gpu.py tests/test_shim_sequence_gpu.py tests/test_errors_gpu.py -q -p no:cacheprovider 2>&1 | tail -40 > $OUT/pytest.log
cat $OUT/pytest.log
timeout 120 python - > $OUT/freeflyer_record.json 2> $OUT/freeflyer.err <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench, __graft_entry__ as g
pkg = g.load_package()
print(json.dumps(bench.freeflyer_discretize_record(pkg)))
PY
cat $OUT/freeflyer_record.json; tail -3 $OUT/freeflyer.err
timeout 480 python tools/starship_n100.py 8 16 $OUT/starship_n100_scvx.json > $OUT/starship_n100.log 2>&1
tail -2 $OUT/starship_n100.log | cut -c1-1800
