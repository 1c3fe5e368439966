"""Which loop is right on the nominal quadrotor instance at hom = 50?  GuSTO `pen = :softplus`, N = 16, Nsub = 10,
12 iterations (tests/test_gusto_gpu.py::test_gusto_softplus_loop_matches_oracle): the oracle loop ends at J_aug = 1.332647, the device
loop at 1.298704.  CPU-only analysis:
  (1) the oracle loop (oracle/gusto_ref.py + oracle/ipm.py::solve_exp), per iteration;
  (2) TEACHER-FORCED: the product's template about the oracle's reference of every iteration, solved by the host build of the
      product's solver (oracle/conic_host.py) -> relative difference of the optimal values (solver parity, no path effects);
  (3) the TWIN loop: the oracle's literal loop with the product's solver behind it -> where the two PATHS part.

    python tools/softplus_forced.py [hom = 50] [instance: 0 nominal | 1 goal + 2 %]
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = {0: "OPTIMAL", 1: "ALMOST_OPTIMAL", 2: "ITERATION_LIMIT", 3: "NUMERICAL_ERROR", 4: "INFEASIBLE", 5: "DUAL_INFEASIBLE"}


def main():
    hom = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
    inst = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import __graft_entry__ as graft
    graft.load_oracle().build()
    pkg = graft.load_package()
    from oracle import conic_host, gusto_ref, ipm, ptr_ref
    from oracle.models import MODELS
    from template_util import make_src, template_matrices
    mdl = MODELS["quadrotor"]()
    op = gusto_ref.quadrotor_test_parameters(16, 10, 12)
    op.pen, op.hom = "softplus", hom
    pp = mdl.nominal_pp().copy()
    if inst == 1:
        pp[6:9] *= 1.02
    st, oh = gusto_ref.gusto_solve("quadrotor", op, pp=pp)
    scale = ptr_ref.Scaling(*mdl.bbox())
    mr = pkg.subproblem.ModelRows(pkg.REGISTRY["quadrotor"]())
    T = pkg.subproblem.build_gusto(mr, 16, scale, pen="softplus", hom=hom)
    print("(1)+(2) oracle loop %s, %d iterations; forced = product template + host solver about the ORACLE's reference" % (st, len(oh)))
    for k, rec in enumerate(oh):
        v, G, A, P = template_matrices(T, make_src(T, mdl, rec["ref"], pp, [rec["eta"], rec["lam"]]))
        r = conic_host._solve(v["c"], G, v["h"], T.l, T.q, A, v["b"], P=P)
        la = r["pcost"] + T.cost_const
        print("  k=%2d lam=%.2e eta=%.3g oracle L_aug=% .9e forced=% .9e rel=%.2e status=%s acc=%s rho=%.4f J_aug=%.6f" % (
            k + 1, rec["lam"], rec["eta"], rec["sub"]["L_aug"], la, abs(la - rec["sub"]["L_aug"]) / max(1.0, abs(rec["sub"]["L_aug"])),
            NAMES[int(r["status"])], rec.get("accept"), rec.get("rho", np.nan), rec["J_aug"]))
    # (3) twin loop
    orig = ipm.solve

    def product_solve(c, G, h, l, q, A=None, b=None, P=None, **kw):
        m = l + sum(q)
        q2 = list(q) + [-3] * ((G.shape[0] - m) // 3)
        r = conic_host._solve(c, G, h, l, q2, A, b, P=sp.triu(sp.csc_matrix(P), format="csc") if P is not None else None)
        return dict(status=NAMES[int(r["status"])], x=r["x"], y=r["y"], z=r["z"], s=r["s"], pcost=float(r["pcost"]),
                    dcost=float(r["dcost"]), gap=float(r["gap"]), pres=float(r["pres"]), dres=float(r["dres"]), iters=int(r["iters"]))
    ipm.solve = product_solve
    ptr_ref.ipm.solve = product_solve
    st2, th = gusto_ref.gusto_solve("quadrotor", op, pp=pp)
    ipm.solve = orig; ptr_ref.ipm.solve = orig
    print("(3) twin loop (oracle loop, product solver) %s, %d iterations" % (st2, len(th)))
    for k, (a, b) in enumerate(zip(oh, th)):
        dref = float(np.abs((a["ref"].xd - b["ref"].xd) / scale.Sx).max())
        print("  k=%2d ref diff (scaled) %.2e  L_aug oracle % .9e twin % .9e  J_aug % .6f / % .6f  rho %.4f / %.4f  accept %s / %s" % (
            k + 1, dref, a["sub"]["L_aug"], b["sub"]["L_aug"], a["J_aug"], b["J_aug"], a.get("rho", np.nan), b.get("rho", np.nan),
            a.get("accept"), b.get("accept")))
    print("end: oracle J_aug %.6f, twin J_aug %.6f" % (oh[-1]["J_aug"], th[-1]["J_aug"]))


if __name__ == "__main__":
    main()
