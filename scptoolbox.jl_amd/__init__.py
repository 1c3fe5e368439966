"""scptoolbox.jl_amd -- MI355X-native SCP inner loop behind the SCPToolbox.jl
solver contract (discretize! + solve_subproblem! under PTR).

The directory name contains a dot, so it cannot be imported with a plain
`import`; use `__graft_entry__.load_package()` (registers it as
`scptoolbox_jl_amd`).
"""
from . import _lib  # noqa: F401
from .models import REGISTRY, NativeModel  # noqa: F401
from .scp import FOH, IMPULSE, DLTV, SCPProblem, SCPScaling, SubproblemSolutionBatch, discretize_, propagate, continuous_time, LinearTrajectory, ImpulseTrajectory, device_guess, device_guess_failures  # noqa: F401
from .problem import TrajectoryProblem  # noqa: F401
from . import ptr as PTR  # noqa: F401
from . import dist  # noqa: F401
from . import conic  # noqa: F401
from . import affine, subproblem, generic  # noqa: F401
from . import scvx as SCvx  # noqa: F401
from . import gusto as GuSTO  # noqa: F401
