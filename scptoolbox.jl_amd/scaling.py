"""`compute_scaling` (src/solvers/scp.jl:376-517): variable bounding boxes from min / max conic programs over the convex
sets X and U, solved as ONE batch on the device (all programs of a set share the constraints; only the cost differs).

The reference builds, for every component without user advice, a fresh ConicProgram with variables (x, u, p), the set's
constraints at every node of the time grid, and the cost +-x_i / +-u_i / +-p_i (scp.jl:430-475); DUAL_INFEASIBLE
(unbounded) and NUMERICAL_ERROR leave the default box [0, 1] (:470-477).  Here the 2 (nx + np) programs over X and the
2 (nu + np) programs over U are two `ConicProgramBatch` solves."""
import numpy as np
import scipy.sparse as sp

from .conic import ALMOST_OPTIMAL, DUAL_INFEASIBLE, ITERATION_LIMIT, NUMERICAL_ERROR, OPTIMAL, ConicProgramBatch
from .scp import SCPScaling


def set_programs(mr, N):
    """(G, h, l, q) over the variables [x; u; p] for the X rows (state/parameter only) and the U rows (any input column)
    of the model at all N nodes; parameter-only rows go to both (they are part of both sets in the reference's models)."""
    nx, nu, np_ = mr.nx, mr.nu, mr.np
    nv = nx + nu + np_
    out = {}
    rows = {"X": dict(lin=[], soc=[]), "U": dict(lin=[], soc=[])}
    for k in range(1, N + 1):
        L, Lp, l, Mm, m = mr.rows(N, k)
        for i in range(mr.nl):
            which = "U" if np.any(L[i, nx:] != 0.0) else "X"
            if not np.any(L[i] != 0.0) and not np.any(Lp[i] != 0.0):
                continue                       # placeholder row (e.g. glide slope at the terminal node)
            rows[which]["lin"].append((np.concatenate([L[i], Lp[i]]), l[i]))
        for c in range(mr.nsoc):
            blk = Mm[4 * c:4 * c + 4]
            which = "U" if np.any(blk[:, nx:] != 0.0) else "X"
            rows[which]["soc"].append((np.hstack([blk, np.zeros((4, np_))]), m[4 * c:4 * c + 4]))
    if mr.ng > 0:
        Lg, lg = mr.global_rows(N)
        for i in range(mr.ng):
            r = (np.concatenate([np.zeros(nx + nu), Lg[i]]), lg[i])
            rows["X"]["lin"].append(r); rows["U"]["lin"].append(r)
    for which in ("X", "U"):
        lin, soc = rows[which]["lin"], rows[which]["soc"]
        # drop exact duplicates (time-invariant sets repeat the same rows at every node)
        seen, ulin = set(), []
        for a, b in lin:
            key = (tuple(np.round(a, 14)), round(float(b), 14))
            if key not in seen:
                seen.add(key); ulin.append((a, b))
        seen, usoc = set(), []
        for a, b in soc:
            key = (tuple(np.round(a.reshape(-1), 14)), tuple(np.round(b, 14)))
            if key not in seen:
                seen.add(key); usoc.append((a, b))
        Gl = np.array([a for a, _ in ulin]).reshape(-1, nv)
        hl = -np.array([b for _, b in ulin])                    # L z + l <= 0  ->  G = L, h = -l
        Gs = -np.vstack([a for a, _ in usoc]) if usoc else np.zeros((0, nv))    # M z + m in Q -> G = -M, h = m
        hs = np.concatenate([b for _, b in usoc]) if usoc else np.zeros(0)
        out[which] = (sp.csc_matrix(np.vstack([Gl, Gs])), np.concatenate([hl, hs]), len(ulin), [4] * len(usoc))
    return out


def compute_scaling(mr, N, advice=None, solver=ConicProgramBatch, **opts):
    """Returns (SCPScaling, dict of bounding boxes and solver statuses).  advice: dict xrg/urg/prg -> list of (min, max)
    or None per component (traj.xrg / urg / prg, src/parser/problem.jl:241-300)."""
    nx, nu, np_ = mr.nx, mr.nu, mr.np
    nv = nx + nu + np_
    advice = advice or {}
    bbox = dict(x=np.tile([0.0, 1.0], (nx, 1)), u=np.tile([0.0, 1.0], (nu, 1)), p=np.tile([0.0, 1.0], (np_, 1)))
    progs = set_programs(mr, N)
    status = {}
    # scp.jl:404-428: (x over X), (u over U), (p over X), (p over U) -- later definitions overwrite earlier ones
    defs = [("x", "X", 0, nx, "xrg"), ("u", "U", nx, nu, "urg"), ("p", "X", nx + nu, np_, "prg"), ("p", "U", nx + nu, np_, "prg")]
    for var, which, off, dim, adv in defs:
        todo = [i for i in range(dim) if advice.get(adv) is None or advice[adv][i] is None]
        for i in range(dim):
            if i not in todo:
                bbox[var][i] = advice[adv][i]
        G, h, l, q = progs[which]
        if not todo or G.shape[0] == 0:
            continue                                  # no constraints: every direction unbounded, box stays [0, 1]
        cs = np.zeros((2 * len(todo), nv))
        for t, i in enumerate(todo):
            cs[2 * t, off + i] = 1.0                  # j = 1: minimise
            cs[2 * t + 1, off + i] = -1.0             # j = 2: maximise
        prog = solver(nv, G, l, q, batch_capacity=cs.shape[0])
        r = prog.solve(cs, h, shared=("h",), **opts)
        if hasattr(prog, "close"):
            prog.close()
        for t, i in enumerate(todo):
            for j in range(2):
                st = int(r["status"][2 * t + j])
                status[(var, which, i, j)] = st
                if st in (OPTIMAL, ALMOST_OPTIMAL):
                    bbox[var][i, j] = (1.0 if j == 0 else -1.0) * r["pcost"][2 * t + j]
                elif st == ITERATION_LIMIT and abs(r["pcost"][2 * t + j]) > 1e4 * (1.0 + np.abs(h).max()):
                    # an unbounded direction whose certificate stalled short of the tolerance (the iterates of a
                    # non-embedded interior-point method diverge along the ray): same outcome as DUAL_INFEASIBLE.  The cost of a
                    # BOUNDED direction is at most of the order of |h|; 1e4 times that is a diverged iterate whatever the summation
                    # order of the factorisation left of it (1.8e8 / 4.7e7 on the rocket's glide-slope cone in two launch geometries)
                    status[(var, which, i, j)] = DUAL_INFEASIBLE
                elif st not in (DUAL_INFEASIBLE, NUMERICAL_ERROR):
                    raise RuntimeError("SCP_SCALING_FAILED: solver status %d for %s[%d]" % (st, var, i))   # scp.jl:470-474
    return SCPScaling(bbox["x"], bbox["u"], bbox["p"]), dict(bbox=bbox, status=status)
