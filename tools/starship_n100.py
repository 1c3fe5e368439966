"""BASELINE.json configs[2] at its stated size, run to the END of the reference's stopping rule (bench.py's
starship_scvx_record with the time budget lifted): python tools/starship_n100.py [batch] [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

pkg = graft.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rec = bench.starship_scvx_record(pkg, B=B, budget_s=float(sys.argv[3]) if len(sys.argv) > 3 else 1e9, detail=True)
s = json.dumps(rec)
print(s)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(s + "\n")
