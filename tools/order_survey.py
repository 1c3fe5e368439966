"""Symbolic statistics (nnz(L), multiply-adds, dissection depth, elimination levels) of the subproblem templates under the
sequential order, the best dissection and the order Engine::create would pick (CONIC_HOST_WORKERS = lanes per problem):
python tools/order_survey.py [quadrotor_gusto quadrotor_scvx rocket_ptr freeflyer_gusto50 freeflyer_gusto200 starship_scvx]
CPU only (oracle/_build/libconic_host.so)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
import conic_host
import scipy.sparse as sp

def stats(T, label):
    # analysis only: B = 0
    for order in ("seq", "nd", "best"):
        os.environ["CONIC_HOST_ORDER"] = order
        t0 = time.perf_counter()
        r = conic_host.analyse(T)
        print("%-28s %-5s n %5d p %5d m %5d  nnzL %8d madds %9d depth %d levels %4d   %.2f s" % (label, order, T.n, T.p, T.m, r[0], r[1], r[4], r[5], time.perf_counter() - t0), flush=True)
        if order != "nd":
            # critical path of the level-scheduled kernel when a worker owns a whole row / a whole pair list: per level
            # max(work / workers, longest item), in multiply-adds; W = 1024 (one problem per workgroup)
            pr = conic_host.schedule_profile(T)
            W = int(os.environ.get("CONIC_HOST_WORKERS", "256"))
            piv = np.maximum(np.ceil(pr[:, 1] / W), pr[:, 2]).sum(); ent = np.maximum(np.ceil(pr[:, 4] / W), pr[:, 5]).sum()
            pivc = np.maximum(np.ceil(pr[:, 1] / W), pr[:, 6]).sum(); entc = np.maximum(np.ceil(pr[:, 4] / W), pr[:, 7]).sum()
            print("%-28s %-5s   critical path per factorisation, whole items: pivots %d (balanced %d), entries %d (balanced %d); per forward "
                  "sweep %d; longest row %d, longest pair list %d | long items chunked: pivots / forward sweep %d, entries %d" % (
                      label, order, piv, np.ceil(pr[:, 1] / W).sum(), ent, np.ceil(pr[:, 4] / W).sum(), piv, pr[:, 2].max(), pr[:, 5].max(), pivc, entc), flush=True)

cases = sys.argv[1:] or ["quadrotor_gusto", "quadrotor_scvx", "rocket_ptr", "freeflyer_gusto50", "freeflyer_scvx50", "starship_scvx", "freeflyer_gusto200"]
for c in cases:
    if c.startswith("quadrotor"):
        name, N = "quadrotor", 30
    elif c.startswith("rocket"):
        name, N = "rocket_landing", 100
    elif c.startswith("freeflyer"):
        name, N = "freeflyer", 200 if c.endswith("200") else 50
    else:
        name, N = "starship", 100
    traj = pkg.TrajectoryProblem(name) if name != "starship" else pkg.TrajectoryProblem(name, hs=1.0)
    pm = traj.mdl

    class Pars: pass
    pars = Pars(); pars.N = N
    if hasattr(pm, "bind"):
        pm.bind(pars)
    from scptoolbox_jl_amd.scp import SCPScaling
    scale = SCPScaling(*pm.scale_advice())
    mr = pkg.subproblem.ModelRows(pm)
    if "gusto" in c:
        T = pkg.subproblem.build_gusto(mr, N, scale)
    elif "scvx" in c:
        T = pkg.subproblem.build_scvx(mr, N, scale, 30.0)
    else:
        T = pkg.subproblem.build_ptr(mr, N, scale, 1e3, 0.1)
    stats(T, c)
