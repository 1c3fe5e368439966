"""tools/conic_reg_sweep.py for the 2 x 15 Starship SCvx N = 100 subproblems of tests/golden/starship_N100_scvx_long{,_t21}.npz (nested order, maxit 1000)."""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
from oracle import conic_host, ptr_ref
from oracle.models import MODELS
from template_util import make_src, template_matrices
import multiprocessing as mp
GOLD="/root/repo/tests/golden/"
def job(a):
    tag, k, reg = a
    os.environ["CONIC_HOST_ORDER"] = "nd"
    g = np.load(GOLD+"starship_N100_scvx_long%s.npz" % tag)
    N, Nsub, hs = int(g["N"]), int(g["Nsub"]), float(g["hs"])
    mdl = MODELS["starship"](N, hs)
    pm = pkg.REGISTRY["starship"](hs=hs); pm.N = N
    mr = pkg.subproblem.ModelRows(pm)
    scale = ptr_ref.Scaling(*mdl.bbox())
    pars = ptr_ref.PTRParameters(N, Nsub, 3, 1e3, 0.1, 0, 0, 5e-3)
    T = pkg.subproblem.build_scvx(mr, N, scale, 5e2)
    ref = ptr_ref.discretize(mdl, pars, scale, g["all_ref_xd"][k], g["all_ref_ud"][k], g["all_ref_p"][k])
    v, G, A, P = template_matrices(T, make_src(T, mdl, ref, mdl.nominal_pp(), float(g["eta"][k])))
    t0 = time.time()
    r = conic_host._solve(v["c"], G, v["h"], T.l, T.q, A, v["b"], P=P, max_iter=1000, **({} if reg is None else dict(reg=reg)))
    rel = abs(r["pcost"] + T.cost_const - g["L_aug"][k]) / max(1.0, abs(g["L_aug"][k]))
    return (tag, reg, k, int(r["status"]), int(r["iters"]), float(rel), int(r["info"][7]), int(r["info"][6]), time.time()-t0)
if __name__ == "__main__":
    ks = list(range(0, 30, 2))
    jobs = [(t, k, rg) for t in ("", "_t21") for rg in (None, 1e-9, 1e-10) for k in ks]
    with mp.Pool(14) as pool:
        res = pool.map(job, jobs, chunksize=1)
    for t in ("", "_t21"):
        for rg in (None, 1e-9, 1e-10):
            rs = [r for r in res if r[0]==t and r[1]==rg]
            st = np.bincount([r[3] for r in rs], minlength=4)
            print("%-5s reg %-6s st %s worst %.1e iters %d (max %d) refinements %d (%.2f/it) nreg %d  sec %.0f  bad %s" % (t, rg, st.tolist(), max(r[5] for r in rs), sum(r[4] for r in rs), max(r[4] for r in rs),
                  sum(r[6] for r in rs), sum(r[6] for r in rs)/sum(r[4] for r in rs), sum(r[7] for r in rs), sum(r[8] for r in rs), [(r[2], r[3], "%.0e" % r[5]) for r in rs if r[5] > 1e-6 or r[3] > 0]), flush=True)
