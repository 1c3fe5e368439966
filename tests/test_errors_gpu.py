"""Error behaviour of the C ABI on a live handle: integer status codes, nothing throws or aborts across the
boundary (mirrors SCPStatus / SCPError, src/utils/globals.jl:34-56); the Python mirror raises ScpError."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pbm(pkg, B=4, **kw):
    traj = pkg.TrajectoryProblem("quadrotor")
    pars = pkg.PTR.Parameters(N=8, Nsub=4, iter_max=2, **kw)
    return traj, pars, pkg.PTR.create(pars, traj, batch_capacity=B)


def test_batch_larger_than_capacity_is_refused(pkg):
    traj, pars, pbm = _pbm(pkg, B=2)
    pp = np.tile(traj.mdl.nominal_pp(), (3, 1))
    with pytest.raises(pkg._lib.ScpError) as e:
        pkg.PTR.solve(pbm, pp)
    assert e.value.code == 6      # SCP_ERR_BATCH_TOO_LARGE
    # the handle stays usable
    sol, _ = pkg.PTR.solve(pbm, pp[:2])
    assert len(sol.status) == 2
    pbm.close()


def test_iterate_before_init_and_null_arguments(pkg):
    traj, pars, pbm = _pbm(pkg)
    L = pkg._lib.lib()
    n = ctypes.c_int(0)
    assert L.scp_ptr_iterate(pbm.handle, ctypes.byref(n)) == 1           # SCP_ERR_BAD_ARGUMENT: no batch initialised
    assert L.scp_ptr_restart(pbm.handle) == 1
    assert L.scp_ptr_init_host(pbm.handle, 1, None, None, None, None, None) == 1
    assert L.scp_discretize_batch_host(pbm.handle, 1, None, None, None, None, None, None, None, None, None, None, None,
                                       None) == 1
    assert L.scp_sync(None) == 1
    assert L.scp_problem_destroy(None) == 1
    pbm.close()


def test_unsupported_options_are_reported_not_ignored(pkg):
    traj = pkg.TrajectoryProblem("quadrotor")
    # trust-region norms 1, 2, 4 run on the generic conic path; anything else is a loud status, never ignored
    pars = pkg.PTR.Parameters(N=8, Nsub=4, iter_max=2, q_tr=3.0)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=1)
    with pytest.raises(ValueError):
        pkg.PTR.solve(pbm, traj.mdl.nominal_pp()[None])
    pbm.close()
    # the structured fast path itself still refuses q_tr != Inf at the C ABI (SCP_ERR_UNSUPPORTED)
    pars = pkg.PTR.Parameters(N=8, Nsub=4, iter_max=2, q_tr=2.0)
    pbm = pkg.PTR.create(pars, traj, batch_capacity=1)
    with pytest.raises(pkg._lib.ScpError) as e:
        pkg.PTR.upload(pbm, traj.mdl.nominal_pp()[None])
    assert e.value.code == 7      # SCP_ERR_UNSUPPORTED
    pbm.close()
    # IMPULSE discretisation needs the model's impulse response: the rocket-landing model has none
    d = pkg._lib.ScpProblemDesc()
    d.model_id = 2
    d.N, d.Nsub, d.batch_capacity, d.disc_method = 8, 4, 1, 1
    h = ctypes.c_void_p()
    assert pkg._lib.lib().scp_problem_create(ctypes.byref(d), ctypes.byref(h)) == 7
