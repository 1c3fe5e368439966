"""GPU debug aid: tiny random SOCPs through the device solver for several launch geometries."""
import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
from test_conic_cpu import random_socp
from oracle import ipm
pkg = graft.load_package()
rng = np.random.default_rng(1)
for trial in range(4):
    q = [(4, 3, 5), (3,), (), (6, 6)][trial % 4]
    c, G, h, l, q, A, b = random_socp(rng, n=10 + trial, pe=trial % 4, l=5 + trial, q=q)
    r0 = ipm.solve(c, G, h, l, q, A, b)
    for B in (1, 64, 65):
        prog = pkg.conic.ConicProgramBatch(c.size, G, l, q, A=A if A.shape[0] else None, batch_capacity=B)
        r1 = prog.solve(np.tile(c, (B, 1)), np.tile(h, (B, 1)), b=np.tile(b, (B, 1)) if A.shape[0] else None)
        print("trial", trial, "B", B, "waves", os.environ.get("SCP_CONIC_WAVES"), "oracle", r0["status"], r0["iters"], "dev status", r1["status"][:3], r1["status"][-1], "iters", r1["iters"][:2], "pcost", r1["pcost"][0], r0["pcost"], "regs", r1["dyn_regs"][0], flush=True)
        prog.close()
