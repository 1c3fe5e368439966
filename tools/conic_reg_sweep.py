"""Static regularisation of the generic conic solver swept on the HOST build (oracle/_build/libconic_host.so) over the teacher-forced
goldens (quadrotor / free-flyer, SCvx / GuSTO, sequential and priced order): statuses, worst optimal-value error, IPM iterations and
refinement steps.  CPU only: OMP_NUM_THREADS=1 python tools/conic_reg_sweep.py   (Starship N = 100: tools/conic_reg_sweep_starship.py)"""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
from oracle import conic_host, ptr_ref
from oracle.models import MODELS
from template_util import make_src, template_matrices, OracleRows
import multiprocessing as mp
GOLD="/root/repo/tests/golden/"
def job(a):
    case, algo, kw, order = a
    os.environ["CONIC_HOST_ORDER"] = order
    if case == "quadrotor":
        N=30; mdl = MODELS["quadrotor"](); mr = pkg.subproblem.ModelRows(pkg.REGISTRY["quadrotor"]()); fc = {}
    else:
        N=50; mdl = MODELS["freeflyer"](N); mr = OracleRows(mdl, N); fc = dict(Fcols=[0])
    scale = ptr_ref.Scaling(*mdl.bbox())
    pars = ptr_ref.PTRParameters(N, 15, 3, 1e3, 0.1, 0, 0, 1e-3)
    g = np.load(GOLD+"teacher_forced_%s_%s_N%d.npz" % (algo, case, N))
    T = (pkg.subproblem.build_scvx(mr, N, scale, 30.0 if case=="quadrotor" else 1e3) if algo=="scvx" else pkg.subproblem.build_gusto(mr, N, scale))
    st = np.zeros(4, int); worst = 0.0; its = 0; nrf = 0; bad = []
    inst = range(8) if case == "freeflyer" else (0, 7, 21, 40, 63, 5, 33, 50)
    for b in inst:
        for k in np.flatnonzero(g["valid"][b]):
            ref = ptr_ref.discretize(mdl, pars, scale, g["ref_xd"][b,k], g["ref_ud"][b,k], g["ref_p"][b,k])
            scal = float(g["eta"][b,k]) if algo=="scvx" else [float(g["eta"][b,k]), float(g["lam"][b,k])]
            v,G,A,P = template_matrices(T, make_src(T, mdl, ref, g["pp"][b], scal, **fc))
            r = conic_host._solve(v["c"], G, v["h"], T.l, T.q, A, v["b"], P=P, **kw)
            rel = abs(r["pcost"]+T.cost_const-g["pcost"][b,k])/max(1,abs(g["pcost"][b,k]))
            st[min(int(r["status"]),3)] += 1; its += int(r["iters"]); nrf += int(r["info"][7])
            if rel > 1e-6: bad.append((b,int(k),int(r["status"]),float("%.1e" % rel)))
            worst = max(worst, rel)
    return "%-10s %-5s %-5s %-28s st %s worst %.1e iters %d refinements %d (%.2f/it) bad %s" % (case, algo, order, kw, st.tolist(), worst, its, nrf, nrf/its, bad[:6])
if __name__ == "__main__":
    kws = [dict(), dict(reg=1e-9), dict(reg=1e-10), dict(reg=1e-11)]
    jobs = [(c,a,kw,o) for o in ("seq","best") for c in ("quadrotor","freeflyer") for a in ("scvx","gusto") for kw in kws]
    with mp.Pool(14) as pool:
        res = pool.map(job, jobs, chunksize=1)
    for r in sorted(res): print(r)
