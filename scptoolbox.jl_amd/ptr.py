"""Host-side mirror of the PTR algorithm interface (src/solvers/ptr.jl)."""
from dataclasses import dataclass, field

from .scp import FOH, SCPProblem


@dataclass
class Parameters:
    """`PTR.Parameters`, src/solvers/ptr.jl:57-71 (same field order)."""
    N: int
    Nsub: int
    iter_max: int
    disc_method: int = FOH
    wvc: float = 1e3
    wtr: float = 0.1
    eps_abs: float = 1e-5   # ε_abs
    eps_rel: float = 1e-4   # ε_rel
    feas_tol: float = 1e-3
    q_tr: float = float("inf")
    q_exit: float = float("inf")
    solver: object = None    # the reference passes a Module (ECOS); here: solver name / None = native ADMM
    solver_opts: dict = field(default_factory=dict)


def create(pars, traj, batch_capacity=1, device=0):
    """`PTR.create(pars, traj)`, src/solvers/ptr.jl:148-195."""
    traj.scp = pars
    return SCPProblem(pars, traj, batch_capacity=batch_capacity, device=device)
