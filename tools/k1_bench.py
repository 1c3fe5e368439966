"""discretize! kernels alone (for the profiler): freeflyer N = 200 batch 4096 (K1x, K1 and the mix) and Starship N = 100 batch 256
(K1 reference form, fp64 and fp32) -- the two sub-records of bench.py that price K1 on state-dependent Jacobians."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

pkg = graft.load_package()
print(json.dumps(dict(freeflyer=bench.freeflyer_discretize_record(pkg), starship=bench.fp32_tolerance_record(pkg))))
