cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04f}
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_starship_gpu.py tests/test_conic_gpu.py tests/test_generic_gpu.py -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
( time timeout 500 python tools/starship_n100.py 256 $OUT/starship_n100_scvx.json 300 ) > $OUT/starship_n100.log 2>&1
python - <<'PY'
import json,sys,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r04f/starship_n100_scvx.json"))
for k,v in d.items():
    if k not in ("nominal","cost_nominal"): print(k, v)
print("nominal eta", d["nominal"]["eta"][-6:], "L", d["nominal"]["L"][-4:])
PY
tail -3 $OUT/starship_n100.log | cut -c1-300
