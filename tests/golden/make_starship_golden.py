"""Golden fixture for the Starship landing flip at the reference's own test configuration
(test/examples/starship_flip/tests.jl:35-49, 77-98: N = 31, Nsub = 100).

    python tests/golden/make_starship_golden.py

starship_N31.npz: the reference's initial guess (bang-bang flip + convex descent, product code
scptoolbox.jl_amd/starship_guess.py with the ORACLE's interior-point solver behind it), the altitude normalisation hs
it sets, and the histories + final trajectories of the oracle's literal PTR loop (15 iterations) and SCvx loop
(lambda = 5e2, eta in [1e-8, 10]) started from it.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
from oracle import ipm, ptr_ref, scvx_ref  # noqa: E402
from oracle.models import MODELS  # noqa: E402

graft.load_package()
from oracle.starship_guess import starship_initial_guess  # noqa: E402


def oracle_batch(c, G0, Gx, hs, l, q, A0, Ax, bs):
    xs, st = [], []
    for t in range(Gx.shape[0]):
        G = sp.csc_matrix((Gx[t], G0.indices, G0.indptr), shape=G0.shape)
        A = sp.csc_matrix((Ax[t], A0.indices, A0.indptr), shape=A0.shape)
        r = ipm.solve(c, G, hs[t], l, q, A, bs[t])
        xs.append(r["x"]); st.append(0 if r["status"] == "OPTIMAL" else (1 if r["status"] == "ALMOST_OPTIMAL" else 2))
        if st[-1] <= 1:
            xs += [r["x"]] * (Gx.shape[0] - t - 1); st += [9] * (Gx.shape[0] - t - 1)   # the reference stops at the first feasible t2
            break
    return np.stack(xs), np.array(st)


def main():
    N, Nsub = 31, 100
    x, u, p, hs = starship_initial_guess(N, oracle_batch)
    mdl = MODELS["starship"](N, hs)
    out = dict(N=N, Nsub=Nsub, guess_x=x, guess_u=u, guess_p=p, hs=hs)
    pars = ptr_ref.PTRParameters(N, Nsub, 15, 1e3, 0.1, 1e-5, 1e-4, 5e-3)
    st, hist = ptr_ref.ptr_solve(mdl, pars, guess=(x, u, p), verbose=True)
    fin = hist[-1]["sol"]
    out.update(ptr_status=st, ptr_iters=len(hist), ptr_J_aug=[h["sub"]["J_aug"] for h in hist], ptr_J=[h["sub"]["J"] for h in hist],
               ptr_feas=[h["sol"].feas for h in hist], ptr_xd=fin.xd, ptr_ud=fin.ud, ptr_p=fin.p,
               ptr_ipm_iters=[h["sub"]["ipm"]["iters"] for h in hist])
    sp_ = scvx_ref.SCvxParameters(N, Nsub, 20, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                  eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3)
    st2, h2 = scvx_ref.scvx_solve(mdl, sp_, guess=(x, u, p), verbose=True)
    fin2 = h2[-1]["sol"]
    out.update(scvx_status=st2, scvx_iters=len(h2), scvx_eta=[h["eta"] for h in h2], scvx_L=[h["sub"]["L"] for h in h2],
               scvx_J_sol=[h.get("J_sol", np.nan) for h in h2], scvx_accept=[bool(h.get("accept", False)) for h in h2],
               scvx_feas=[h["sol"].feas for h in h2], scvx_xd=fin2.xd, scvx_ud=fin2.ud, scvx_p=fin2.p)
    np.savez_compressed(os.path.join(HERE, "starship_N31.npz"), **out)
    print("PTR", st, len(hist), "SCvx", st2, len(h2))


if __name__ == "__main__":
    main()
