"""Outcomes of the ORACLE's literal GuSTO loop on the Monte-Carlo instances of bench.py's `freeflyer_gusto.full_run_reference_grid`
record (free-flyer, reference test parameters freeflyer/tests.jl:84-140, N = 50, Nsub = 15, 15 iterations, initial / terminal
positions +-3 mm, seed = instance index): status, dynamic feasibility, accepted steps, the cost L of the last subproblem.

    OMP_NUM_THREADS=1 python tests/golden/make_freeflyer_gusto_outcomes.py [instances = 128] [processes = 15]
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def run(i):
    from oracle import gusto_ref
    from oracle.models import MODELS
    N = 50
    mdl = MODELS["freeflyer"](N)
    rng = np.random.default_rng(i)
    q = mdl.nominal_pp().copy()
    q[0:3] += 0.003 * rng.uniform(-1, 1, 3); q[13:16] += 0.003 * rng.uniform(-1, 1, 3)        # bench.py, freeflyer_gusto_record.pps
    gp = gusto_ref.GuSTOParameters(N, 15, 15, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0, beta_gr=2.0,
                                   gamma_fail=5.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16, eps_abs=0.0,
                                   eps_rel=0.0, feas_tol=1e-3)
    st, h = gusto_ref.gusto_solve(mdl, gp, pp=q)
    return (i, 0 if st.split()[0] == "SCP_SOLVED" else 1, len(h), bool(h[-1]["sol"].feas), sum(bool(r.get("accept", False)) for r in h),
            float(h[-1]["sub"]["L"]), float(h[-1]["lam"]))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    with mp.Pool(procs) as pool:
        res = pool.map(run, range(n), chunksize=1)
    res.sort()
    np.savez_compressed(os.path.join(HERE, "gusto_outcomes_freeflyer_N50.npz"), status=np.array([r[1] for r in res], np.int8),
                        iterations=np.array([r[2] for r in res], np.int16), feas=np.array([r[3] for r in res]),
                        accepted=np.array([r[4] for r in res], np.int16), L_last=np.array([r[5] for r in res]),
                        lam_last=np.array([r[6] for r in res]), N=50, Nsub=15, iter_max=15)
    print("solved %.4f feasible %.4f median L %.6f" % (np.mean([r[1] == 0 for r in res]), np.mean([r[3] for r in res]), np.median([r[5] for r in res])))


if __name__ == "__main__":
    main()
