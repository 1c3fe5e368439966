"""Outcomes of the ORACLE's literal GuSTO loop on the Monte-Carlo instances of bench.py's `freeflyer_gusto.full_run_reference_grid`
record (free-flyer, reference test parameters freeflyer/tests.jl:84-140, N = 50, Nsub = 15, 15 iterations, initial / terminal
positions +-3 mm, seed = instance index): status, dynamic feasibility and the per-iteration record of every instance (cost L of
the subproblem, accept / reject, radius, penalty weight, stop flag).

With eps_abs = eps_rel = 0 the reference's rule `dJ <= eps_rel or deviation <= eps_abs` (gusto.jl:1203-1230) fires only on EXACT
equality: a loop that has reached its fixed point to the last bit (J_aug of the solution identical to its reference's) stops
there -- 22 of the 128 oracle loops do so after 10-14 iterations.  Whether the last bits coincide is round-off, so a correct
device loop may run on to iteration 15 on the same instance: comparisons therefore use each loop's OWN last iteration and the
iterations both loops executed (tests/test_outcomes_gpu.py, bench.py).

    OMP_NUM_THREADS=1 python tests/golden/make_freeflyer_gusto_outcomes.py [instances = 128] [processes = 7]
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
ITERS = 15


def run(i):
    from oracle import gusto_ref
    from oracle.models import MODELS
    N = 50
    mdl = MODELS["freeflyer"](N)
    rng = np.random.default_rng(i)
    q = mdl.nominal_pp().copy()
    q[0:3] += 0.003 * rng.uniform(-1, 1, 3); q[13:16] += 0.003 * rng.uniform(-1, 1, 3)        # bench.py, freeflyer_gusto_record.pps
    gp = gusto_ref.GuSTOParameters(N, 15, ITERS, lam_init=1e4, lam_max=1e9, rho_0=0.1, rho_1=0.5, beta_sh=2.0, beta_gr=2.0,
                                   gamma_fail=5.0, eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, mu=0.8, iter_mu=16, eps_abs=0.0,
                                   eps_rel=0.0, feas_tol=1e-3)
    st, h = gusto_ref.gusto_solve(mdl, gp, pp=q)
    L = np.full(ITERS, np.nan); J = np.full(ITERS, np.nan); eta = np.full(ITERS, np.nan); lam = np.full(ITERS, np.nan)
    acc = np.full(ITERS, -1, np.int8)            # -1: no decision at that iteration (stopped there / never reached)
    for k, r in enumerate(h):
        L[k] = r["sub"]["L"]; J[k] = r.get("J_aug", np.nan); eta[k] = r["eta"]; lam[k] = r["lam"]
        if "accept" in r:
            acc[k] = int(r["accept"])
    return (i, 0 if st.split()[0] == "SCP_SOLVED" else 1, len(h), bool(h[-1]["sol"].feas), bool(h[-1]["stop"]), L, J, eta, lam, acc)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    with mp.Pool(procs) as pool:
        res = pool.map(run, range(n), chunksize=1)
    res.sort(key=lambda r: r[0])
    acc = np.stack([r[9] for r in res]); L = np.stack([r[5] for r in res]); its = np.array([r[2] for r in res], np.int16)
    np.savez_compressed(os.path.join(HERE, "gusto_outcomes_freeflyer_N50.npz"), status=np.array([r[1] for r in res], np.int8),
                        iterations=its, feas=np.array([r[3] for r in res]), stopped=np.array([r[4] for r in res]),
                        accepted=(acc == 1).sum(axis=1).astype(np.int16), accept=acc, L=L, J_aug=np.stack([r[6] for r in res]),
                        eta=np.stack([r[7] for r in res]), lam=np.stack([r[8] for r in res]),
                        L_last=L[np.arange(len(res)), its - 1], lam_last=np.stack([r[8] for r in res])[np.arange(len(res)), its - 1],
                        N=50, Nsub=15, iter_max=ITERS)
    print("solved %.4f feasible %.4f stopped early %d median L %.6f" % (np.mean([r[1] == 0 for r in res]), np.mean([r[3] for r in res]),
                                                                        int((its < ITERS).sum()), np.median(L[np.arange(len(res)), its - 1])))


if __name__ == "__main__":
    main()
