"""CPU twin of the device SCvx loop on BASELINE.json configs[2] at its stated size (Starship, N = 100, Nsub = 100): the oracle's
literal loop (oracle/scvx_ref.py) with the PRODUCT's conic solver (host build, nested order) behind it, from the golden's guess,
compared iteration by iteration with the oracle's own record (tests/golden/starship_N100_scvx_long.npz).  Predicts the outcome of
tests/test_starship_gpu.py::test_scvx_thirty_iterations_at_config_size_follow_the_oracle before a GPU is spent on it.

    OMP_NUM_THREADS=4 python tools/starship_twin.py [iterations = 30] [tag = '' | _t21]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    tag = sys.argv[2] if len(sys.argv) > 2 else ""        # "" = the oracle's record from the 20 s guess, "_t21" = from the 21 s guess
    import __graft_entry__ as graft
    graft.load_oracle().build()
    import scipy.sparse as sp  # noqa: F401
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from template_util import make_src, template_matrices
    from oracle import conic_host, ptr_ref
    pkg = graft.load_package()
    os.environ["CONIC_HOST_ORDER"] = "nd"
    NAMES = {0: "OPTIMAL", 1: "ALMOST_OPTIMAL", 2: "ITERATION_LIMIT", 3: "NUMERICAL_ERROR", 4: "INFEASIBLE", 5: "DUAL_INFEASIBLE"}
    cache = {}

    def product_subproblem(mdl, pars, scale, ref, pp, ipm_opts=None, algo="ptr", eta=None):
        """the PRODUCT's formulation (row-equilibrated template, subproblem.py) + the product's solver (host build, max_iter = 1000)"""
        assert algo == "scvx"
        if "T" not in cache:
            pm = pkg.REGISTRY["starship"](hs=mdl.hs) if hasattr(mdl, "hs") else pkg.REGISTRY["starship"]()
            pm.N = pars.N
            cache["T"] = pkg.subproblem.build_scvx(pkg.subproblem.ModelRows(pm), pars.N, scale, pars.lam)
        T = cache["T"]
        v, G, A, P = template_matrices(T, make_src(T, mdl, ref, pp, float(eta)))
        r = conic_host._solve(v["c"], G, v["h"], T.l, T.q, A, v["b"], P=P, max_iter=1000)
        z = r["x"]
        N = pars.N
        x = np.stack([scale.Sx * z[i] + scale.cx for i in T.variables["xh"].reshape(N, -1)])
        u = np.stack([scale.Su * z[i] + scale.cu for i in T.variables["uh"].reshape(N, -1)])
        p = scale.Sp * z[T.variables["ph"]] + scale.cp
        from oracle import scvx_ref as sr
        L = sr.compute_original_cost(mdl, pars, x, u, p)
        La = float(r["pcost"] + T.cost_const)
        return dict(x=x, u=u, p=p, status=NAMES[int(r["status"])], L=L, L_pen=La - L, L_aug=La, ipm=dict(iters=int(r["iters"])))
    ptr_ref.solve_subproblem = product_subproblem
    from oracle import scvx_ref
    from oracle.models import MODELS
    G = os.path.join(ROOT, "tests", "golden")
    g3 = np.load(os.path.join(G, "starship_N100_scvx3%s.npz" % tag)); g = np.load(os.path.join(G, "starship_N100_scvx_long%s.npz" % tag))
    N, Nsub, hs = int(g["N"]), int(g["Nsub"]), float(g["hs"])
    mdl = MODELS["starship"](N, hs)
    sp_ = scvx_ref.SCvxParameters(N, Nsub, iters, lam=5e2, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0, eta_init=1.0,
                                  eta_lb=1e-8, eta_ub=10.0, eps_abs=1e-5, eps_rel=1e-4, feas_tol=5e-3)
    st, h = scvx_ref.scvx_solve(mdl, sp_, guess=(g3["guess_x"], g3["guess_u"], g3["guess_p"]), verbose=True)
    print(st, len(h))
    for k, r in enumerate(h):
        L, J = r["sub"]["L"], r.get("J_sol", np.nan)
        print("k=%2d eta %.6g/%.6g accept %s/%s  L rel %.2e  J_sol rel %.2e  rho %.4f/%.4f" % (
            k + 1, r["eta"], g["eta"][k], r.get("accept"), bool(g["accept"][k]), abs(L - g["L"][k]) / max(1, abs(g["L"][k])),
            abs(J - g["J_sol"][k]) / max(1, abs(g["J_sol"][k])), r.get("rho", np.nan), g["rho"][k]))


if __name__ == "__main__":
    main()
