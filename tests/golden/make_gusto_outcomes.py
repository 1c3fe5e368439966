"""Outcomes of the ORACLE's literal GuSTO loop (oracle/gusto_ref.py) on the Monte-Carlo instances of bench.py's
`gusto_quadrotor` record (quadrotor, reference test parameters quadrotor/tests.jl:86-130, N = 30, Nsub = 15, 6 iterations,
goal position +-10 %, seed = instance index): status, iterations and the per-iteration record (L_aug, accept / reject, eta,
lambda) of every instance.

Instances whose loop ends SCP_FAILED are re-run with the oracle solver's objective NORMALISED (oracle/ipm.py
`normalise_objective`, the arithmetic of the product's solver once a cost coefficient exceeds 1e4): `fail_sub_status` names the
exit of the failing subproblem (a solver exit: ITERATION_LIMIT / NUMERICAL_ERROR at lambda >= 1e6), `status_normalised` the
loop's outcome with the normalised solver.  tests/test_outcomes_cpu.py asserts that every SCP_FAILED here is such a solver exit
and that the normalised loop solves the instance -- i.e. the device's 100 % SCP_SOLVED on this batch is the algorithm's
outcome and the oracle's 95.7 % is its un-normalised solver giving up.

    python tests/golden/make_gusto_outcomes.py [instances = 1024] [processes = 7]      # ~2 s per instance
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
ITERS = 6
SUB_CODES = {"OPTIMAL": 0, "ALMOST_OPTIMAL": 1, "ITERATION_LIMIT": 2, "NUMERICAL_ERROR": 3}


def one(pp, normalise):
    from oracle import gusto_ref
    op = gusto_ref.quadrotor_test_parameters(30, 15, ITERS)
    op.eps_abs = op.eps_rel = 0.0
    st, oh = gusto_ref.gusto_solve("quadrotor", op, pp=pp, ipm_opts=dict(normalise_objective=True) if normalise else None)
    La = np.full(ITERS, np.nan); eta = np.full(ITERS, np.nan); lam = np.full(ITERS, np.nan); acc = np.full(ITERS, -1, np.int8)
    rho = np.full(ITERS, np.nan)
    for k, r in enumerate(oh):
        La[k] = r["sub"]["L_aug"]; eta[k] = r["eta"]; lam[k] = r["lam"]
        if "accept" in r:
            acc[k] = int(r["accept"]); rho[k] = r["rho"]
    last = oh[-1]
    return (0 if st.split()[0] == "SCP_SOLVED" else 1, len(oh), float(last.get("J_aug", np.nan)), float(last["lam"]),
            SUB_CODES.get(str(last["sub"]["status"]), 9), La, eta, lam, acc, bool(last["sol"].feas), rho)


def run(b):
    os.environ["OMP_NUM_THREADS"] = "1"
    import bench
    from oracle.models import MODELS
    mdl = MODELS["quadrotor"]()
    pp = bench.mc_pp(mdl, 1, b)[0]          # seed = instance index
    r = one(pp, False)
    rn = one(pp, True) if r[0] != 0 else None
    return b, r, rn


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    with mp.Pool(procs) as pool:
        res = pool.map(run, range(n), chunksize=4)
    res.sort(key=lambda r: r[0])
    status = np.array([r[1][0] for r in res], np.int8); iters = np.array([r[1][1] for r in res], np.int16)
    np.savez_compressed(os.path.join(HERE, "gusto_outcomes_quadrotor_N30.npz"), status=status, iterations=iters,
                        J_aug=np.array([r[1][2] for r in res]), lam=np.array([r[1][3] for r in res]),
                        fail_sub_status=np.array([r[1][4] if r[1][0] != 0 else -1 for r in res], np.int8),
                        status_normalised=np.array([r[2][0] if r[2] is not None else r[1][0] for r in res], np.int8),
                        iterations_normalised=np.array([r[2][1] if r[2] is not None else r[1][1] for r in res], np.int16),
                        L_aug=np.stack([r[1][5] for r in res]), eta=np.stack([r[1][6] for r in res]), lam_it=np.stack([r[1][7] for r in res]),
                        accept=np.stack([r[1][8] for r in res]), rho=np.stack([r[1][10] for r in res]), feas=np.array([r[1][9] for r in res]), N=30, Nsub=15, iter_max=ITERS)
    print("solved fraction %.4f of %d; with the normalised solver %.4f" % (
        (status == 0).mean(), n, np.mean([(r[2][0] if r[2] is not None else r[1][0]) == 0 for r in res])))


if __name__ == "__main__":
    main()
