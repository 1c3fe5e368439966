"""CPU twin of the device GuSTO / SCvx / PTR loops: the ORACLE's literal loop (oracle/{gusto,scvx,ptr}_ref.py) with the PRODUCT's
conic solver (host build, oracle/_build/libconic_host.so) behind it instead of oracle/ipm.py.  Predicts what the device loop
does with a solver change before a GPU is available: without the objective normalisation of conic_ipm.hpp the twin ends
54 % of bench.py's first 128 quadrotor GuSTO instances SCP_SOLVED (device, measured on 1 024: 59 %), with it 100 %.

    python tools/cpu_twin.py gusto_quadrotor [instances = 128] [processes = 14]
"""
import multiprocessing as mp
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "OPTIMAL", 1: "ALMOST_OPTIMAL", 2: "ITERATION_LIMIT", 3: "NUMERICAL_ERROR", 4: "INFEASIBLE", 5: "DUAL_INFEASIBLE"}


def install(order="best"):
    """route every conic solve of the oracle loops through the product's solver"""
    from oracle import conic_host, ipm, ptr_ref
    os.environ.setdefault("CONIC_HOST_ORDER", order)

    def product_solve(c, G, h, l, q, A=None, b=None, P=None, **kw):
        r = conic_host._solve(c, G, h, l, q, A, b, P=sp.triu(sp.csc_matrix(P), format="csc") if P is not None else None)
        return dict(status=NAMES[int(r["status"])], x=r["x"], y=r["y"], z=r["z"], s=r["s"], pcost=float(r["pcost"]),
                    dcost=float(r["dcost"]), gap=float(r["gap"]), pres=float(r["pres"]), dres=float(r["dres"]), iters=int(r["iters"]))
    ipm.solve = product_solve
    ptr_ref.ipm.solve = product_solve


def _gusto_quadrotor(b):
    install()
    import bench
    from oracle import gusto_ref
    from oracle.models import MODELS
    pp = bench.mc_pp(MODELS["quadrotor"](), 1, b)[0]
    op = gusto_ref.quadrotor_test_parameters(30, 15, 6)
    op.eps_abs = op.eps_rel = 0.0
    st, oh = gusto_ref.gusto_solve("quadrotor", op, pp=pp)
    return b, st, len(oh), float(oh[-1]["lam"])


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "gusto_quadrotor"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    assert what == "gusto_quadrotor"
    os.environ["OMP_NUM_THREADS"] = "1"
    with mp.Pool(procs) as pool:
        res = pool.map(_gusto_quadrotor, range(n))
    g = np.load(os.path.join(ROOT, "tests", "golden", "gusto_outcomes_quadrotor_N30.npz"))
    ok = np.array([r[1].split()[0] == "SCP_SOLVED" for r in res])
    print("twin solved %.4f ; oracle solved %.4f ; same status %.4f" % (ok.mean(), (g["status"][:n] == 0).mean(), (ok == (g["status"][:n] == 0)).mean()))


if __name__ == "__main__":
    main()
