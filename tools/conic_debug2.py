"""GPU debug aid: the degenerate LP (trial 2 of tests/test_conic_gpu.py) for several regularisations."""
import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
from test_conic_cpu import random_socp
pkg = graft.load_package()
rng = np.random.default_rng(1)
for trial in range(3):
    q = [(4, 3, 5), (3,), (), (6, 6)][trial % 4]
    c, G, h, l, q, A, b = random_socp(rng, n=10 + trial, pe=trial % 4, l=5 + trial, q=q)
    P = sp.diags(rng.uniform(0.1, 1.0, c.size)) if trial % 2 else None
prog = pkg.conic.ConicProgramBatch(c.size, G, l, q, A=A, batch_capacity=1)
for kw in [dict(), dict(reg=1e-8), dict(reg=1e-7), dict(reg=2e-7), dict(reg=1e-10), dict(dyn_delta=1e-6), dict(dyn_eps=1e-10), dict(reg=1e-8, dyn_eps=1e-10), dict(max_iter=4), dict(max_iter=5)]:
    r1 = prog.solve(c[None], h[None], b=b[None], **kw)
    print(kw, r1["status"], r1["iters"], r1["dyn_regs"], r1["refinements"], r1["pcost"], r1["pres"], r1["dres"], r1["gap"], flush=True)
