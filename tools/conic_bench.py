"""GPU timing of the generic conic solver on the golden PTR conic programs (tests/golden/conic_*.npz) replicated to a
batch:  python tools/conic_bench.py [name] [B ...]   -> one JSON line per batch size."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "conic_rocket_landing_N100"
Bs = [int(a) for a in sys.argv[2:]] or [1024, 4096]
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
n, l, q = int(g["n"]), int(g["l"]), list(g["q"])
m = l + sum(q); p = g["b"].shape[1]
G = sp.csc_matrix((np.ones(len(g["Gi"])), g["Gi"], g["Gp"]), shape=(m, n))
A = sp.csc_matrix((np.ones(len(g["Ai"])), g["Ai"], g["Ap"]), shape=(p, n))
P = sp.csc_matrix((np.ones(len(g["Pi"])), g["Pi"], g["Pp"]), shape=(n, n))
opts = {}
for kv in os.environ.get("CONIC_OPTS", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        opts[k] = int(v) if k in ("max_iter", "nref") else float(v)
for B in Bs:
    t0 = time.time()
    prog = pkg.conic.ConicProgramBatch(n, G, l, q, A=A, P=P, batch_capacity=B)
    t_create = time.time() - t0
    tile = lambda a: np.tile(a[1], (B, 1))
    args = dict(b=g["b"][1], Gx=g["Gx"][1], Ax=g["Ax"][1], Px=g["Px"][1], shared=("b", "Gx", "Ax", "Px"))
    best = None
    for rep in range(2):
        r = prog.solve(tile(g["c"]), tile(g["h"]), **args, **opts)
        best = r["seconds"] if best is None else min(best, r["seconds"])
    st = prog.stats()
    prog.close()
    its = r["iters"].astype(float)
    # bytes moved per problem: factor (2 loads / pair) + solves (4 nnzL per solve_raw) -- the algorithmic traffic
    nsolve = 2 * its.mean() + 1 + r["refinements"].mean()
    byt = 8.0 * (its.mean() + 1) * 2 * st["factor_madds"] + 8.0 * nsolve * 4 * st["nnzL"]
    print(json.dumps(dict(name=name, B=B, seconds=best, create_s=round(t_create, 2), solved=float((r["status"] == 0).mean()),
                          iters_mean=its.mean(), refinements_mean=r["refinements"].mean(), stats=st,
                          problems_per_s=B / best, alg_GB_per_problem=byt / 1e9, alg_TBps=byt * B / best / 1e12)), flush=True)
