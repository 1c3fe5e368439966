"""Host-side mirror of `TrajectoryProblem` (src/parser/problem.jl:64-121) for
compiled models.  The reference's closure setters (`problem_set_dynamics!` ...)
have no counterpart: the dynamics, constraints, boundary conditions and cost of
a native model live in the HIP library."""
import numpy as np

from .models import REGISTRY, NativeModel


class TrajectoryProblem:
    """`TrajectoryProblem(mdl)` (src/parser/problem.jl:175-239).  `mdl` is a
    NativeModel instance or the registry name of one."""

    def __init__(self, mdl, **overrides):
        if isinstance(mdl, str):
            mdl = REGISTRY[mdl](**overrides)
        assert isinstance(mdl, NativeModel)
        self.mdl = mdl
        self.nx, self.nu = mdl.nx, mdl.nu
        self.scp = None  # set by SCPProblem (scp.jl:97)

    @property
    def np(self):
        """parameter count; for a model with per-node parameters (free-flyer: 1 + 6 N) known once the grid is (mdl.bind)"""
        return self.mdl.np

    def guess(self, N, pp=None):
        """`traj.guess(N)` (problem.jl:319-322); pp = per-problem data or None."""
        pp = self.mdl.nominal_pp() if pp is None else np.asarray(pp, dtype=np.float64)
        return self.mdl.guess(N, pp)
