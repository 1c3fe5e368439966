"""oracle/cpu_ptr.cpp (the C++/OpenMP restatement of the structured PTR iteration: bench.py's CPU baseline and the sibling the
device results are compared with at the headline size, tests/test_config_size_gpu.py) pinned against the LITERAL loop of
oracle/ptr_ref.py -- every subproblem a literal conic program solved by oracle/ipm.py -- on the first instances of the headline
Monte-Carlo batch (rocket landing, N = 100, Nsub = 15, 15 iterations; tests/golden/ptr_outcomes_rocket_landing_N100.npz)."""
import os

import numpy as np

import bench
from oracle import cpu_ptr
from oracle.models import MODELS


def test_structured_port_equals_the_literal_loop_on_headline_instances():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ptr_outcomes_rocket_landing_N100.npz"))
    nb = 64
    mdl = MODELS["rocket_landing"]()
    r = cpu_ptr.solve_batch("rocket_landing", int(g["N"]), int(g["Nsub"]), int(g["iter_max"]), bench.mc_pp(mdl, nb, 0), threads=0,
                            want_hist=True)

    class Sol:
        pass
    sol = Sol()
    sol.status = ["SCP_SOLVED" if r["hist"][b, :, 5].max() <= 1 else "SCP_FAILED" for b in range(nb)]
    sol.feas = r["stats"][:, 2] > 0
    sol.J_aug = r["hist"][:, -1, 0]
    sol.p = r["p"]
    o = bench.oracle_outcomes_ptr("rocket_landing", int(g["N"]), int(g["Nsub"]), int(g["iter_max"]), 0, sol)     # what bench.py reports
    assert o["instances"] == nb and o["same_status"] == 1.0 and o["same_feasibility_flag"] == 1.0
    assert o["converged_in_both"] >= 55
    # measured on the first 256 instances: median 5.7e-9, maximum 9.4e-7 (costs), 5e-4 s (final time)
    assert o["J_aug_rel_diff_max"] <= 2e-6 and o["tf_abs_diff_max_s"] <= 2e-3
    # the oracle's batch statistics themselves (all 256): every subproblem OPTIMAL, 93.75 % dynamically feasible after 15 iterations
    assert g["ipm_all_optimal"].all() and (g["status"] == 0).all() and abs(g["feas"].mean() - 0.9375) < 1e-12
