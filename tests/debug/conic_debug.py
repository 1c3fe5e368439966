"""GPU debug aid: the random SOCPs of tests/test_conic_gpu.py through the device solver, all launch geometries."""
import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
from test_conic_cpu import random_socp
from oracle import ipm, conic_host
pkg = graft.load_package()
rng = np.random.default_rng(1)
for trial in range(6):
    q = [(4, 3, 5), (3,), (), (6, 6)][trial % 4]
    c, G, h, l, q, A, b = random_socp(rng, n=10 + trial, pe=trial % 4, l=5 + trial, q=q)
    P = sp.diags(rng.uniform(0.1, 1.0, c.size)) if trial % 2 else None
    r0 = ipm.solve(c, G, h, l, q, A, b, P=P)
    rh = conic_host.solve(c, G, h, l, q, A if A.shape[0] else None, b, P=P)
    for B in (1, 65):
        prog = pkg.conic.ConicProgramBatch(c.size, G, l, q, A=A if A.shape[0] else None, P=P, batch_capacity=B)
        r1 = prog.solve(np.tile(c, (B, 1)), np.tile(h, (B, 1)), b=np.tile(b, (B, 1)) if A.shape[0] else None)
        print("trial", trial, "P" if P is not None else "-", "B", B, "waves", os.environ.get("SCP_CONIC_WAVES"), "oracle", r0["status"], r0["iters"], "host", rh["status"], rh["iters"],
              "dev status", r1["status"][:2], r1["status"][-1], "iters", r1["iters"][:2], "pcost", r1["pcost"][0], r0["pcost"], "pres %.1e dres %.1e gap %.1e" % (r1["pres"][0], r1["dres"][0], r1["gap"][0]), "regs", r1["dyn_regs"][0], r1["refinements"][0], flush=True)
        prog.close()
