"""Golden fixtures for the generic conic solver (include/scp_conic.h): the literal PTR conic programs the oracle
(oracle/ptr_ref.py, the restatement of src/solvers/ptr.jl:213-293 + MOI's NormInf/NormOne bridges) hands to its
interior-point solver at the CONFIG sizes, with the oracle's solution.

    python tests/golden/make_conic_golden.py

conic_<model>_N<N>.npz: for three subproblems (first / mid-run / late) of the nominal PTR run: pattern (CSC of P upper,
A, G; l, q), values (c, b, h, Gx, Ax, Px) and the oracle IPM's x, pcost, iteration count.  The patterns of the three
programs are made identical (union, explicit zeros) so that they form one batch.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ipm, ptr_ref  # noqa: E402

CASES = [("quadrotor", 50, 15, 8, (0, 3, 7)), ("rocket_landing", 100, 15, 8, (0, 3, 7))]


def union_pattern(mats):
    """values of every matrix on the union pattern (explicit zeros where absent)."""
    pat = None
    for M in mats:
        Z = sp.csc_matrix(M, copy=True)
        Z.data[:] = 1.0
        pat = Z if pat is None else pat + Z
    pat = sp.csc_matrix(pat); pat.sum_duplicates(); pat.sort_indices()
    pat.data[:] = 1.0
    pc = pat.tocoo()      # entries of a canonical CSC matrix come out in CSC (column-major) order
    lut = {(r, c): k for k, (r, c) in enumerate(zip(pc.row, pc.col))}
    vals = []
    for M in mats:
        Mc = sp.csc_matrix(M).tocoo()
        out = np.zeros(pat.nnz)
        for r, c, v in zip(Mc.row, Mc.col, Mc.data):
            out[lut[(r, c)]] += v
        vals.append(out)
    return pat, vals


def main():
    cap = []
    orig = ipm.solve

    def hook(c, G, h, l, q, A=None, b=None, P=None, **kw):
        r = orig(c, G, h, l, q, A=A, b=b, P=P, **kw)
        cap.append((c, G, h, l, q, A, b, P, r))
        return r
    ptr_ref.ipm.solve = hook
    for model, N, Nsub, iters, picks in CASES:
        cap.clear()
        pars = ptr_ref.PTRParameters(N, Nsub, iters, 1e3, 0.1, 0, 0, 1e-3)
        ptr_ref.ptr_solve(model, pars)
        sel = [cap[i] for i in picks]
        n = sel[0][0].size
        Gp, Gv = union_pattern([s[1] for s in sel])
        Ap, Av = union_pattern([s[5] for s in sel])
        Pp, Pv = union_pattern([sp.triu(sp.csc_matrix(s[7])) for s in sel])
        l, q = sel[0][3], sel[0][4]
        assert all(s[3] == l and list(s[4]) == list(q) for s in sel)
        out = dict(n=n, l=l, q=np.asarray(q, np.int32),
                   Gp=Gp.indptr.astype(np.int32), Gi=Gp.indices.astype(np.int32), Gx=np.stack(Gv),
                   Ap=Ap.indptr.astype(np.int32), Ai=Ap.indices.astype(np.int32), Ax=np.stack(Av),
                   Pp=Pp.indptr.astype(np.int32), Pi=Pp.indices.astype(np.int32), Px=np.stack(Pv),
                   c=np.stack([s[0] for s in sel]), h=np.stack([s[2] for s in sel]), b=np.stack([s[6] for s in sel]),
                   x=np.stack([s[8]["x"] for s in sel]), pcost=np.array([s[8]["pcost"] for s in sel]),
                   iters=np.array([s[8]["iters"] for s in sel]), status=np.array([s[8]["status"] for s in sel]))
        np.savez_compressed(os.path.join(HERE, "conic_%s_N%d.npz" % (model, N)), **out)
        print(model, N, "n", n, "p", Ap.shape[0], "m", Gp.shape[0], "iters", out["iters"], out["status"])


if __name__ == "__main__":
    main()
