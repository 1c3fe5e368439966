"""N > 1 path with the REAL library (-m gpu): world_size-2 gloo processes, both on GPU 0, each running the product's
PTR.solve / SCvx.solve on its contiguous shard of a 64-problem Monte-Carlo batch with the per-iteration convergence
all-reduce (scptoolbox.jl_amd/dist.py) and the stopping criterion ENABLED -- the concatenated result must be bit-identical
to the single-process run of the whole batch, and both ranks must have made the same number of iterate calls (= the
iterations of the slowest problem anywhere in the batch).  RCCL is exercised by the driver's multi-GPU bench; this test
pins the sharding + lockstep logic on the device path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B_TOTAL = 64


def _pp(pkg, model, n):
    mdl = pkg.REGISTRY[model]()
    out = []
    for i in range(n):
        rng = np.random.default_rng(100 + i)
        q = mdl.nominal_pp().copy()
        if model == "quadrotor":
            q[6:9] *= 1 + 0.03 * rng.uniform(-1, 1, 3)
        else:
            q *= 1 + 0.05 * rng.uniform(-1, 1, q.size)
        out.append(q)
    return np.stack(out)


def _solve(pkg, algo, pp, all_reduce):
    if algo == "ptr":
        traj = pkg.TrajectoryProblem("rocket_landing")
        pars = pkg.PTR.Parameters(N=20, Nsub=8, iter_max=14, wvc=1e3, wtr=0.1, eps_abs=1e-4, eps_rel=1e-5, feas_tol=1e-3)
        pbm = pkg.PTR.create(pars, traj, batch_capacity=pp.shape[0])
        sol, hist = pkg.PTR.solve(pbm, pp, all_reduce=all_reduce)
    else:
        traj = pkg.TrajectoryProblem("quadrotor")
        pars = pkg.SCvx.Parameters(N=16, Nsub=8, iter_max=12, lam=30.0, rho_0=0.0, rho_1=0.1, rho_2=0.7, beta_sh=2.0, beta_gr=2.0,
                                   eta_init=1.0, eta_lb=1e-3, eta_ub=10.0, eps_abs=1e-4, eps_rel=1e-3)
        pbm = pkg.SCvx.create(pars, traj, batch_capacity=pp.shape[0])
        sol, hist = pkg.SCvx.solve(pbm, pp, all_reduce=all_reduce)
    pbm.close()
    return sol.xd, sol.ud, sol.p, np.asarray(sol.iterations)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for algo, model in (("ptr", "rocket_landing"), ("scvx", "quadrotor")):
        pp = _pp(pkg, model, B_TOTAL)
        lo, hi = pkg.dist.shard_range(B_TOTAL, rank, world)
        inner = pkg.dist.make_all_reduce(dist)
        calls = [0]

        def ar(n, inner=inner, calls=calls):
            calls[0] += 1
            return inner(n)
        xd, ud, p, its = _solve(pkg, algo, pp[lo:hi], ar)
        res.update({algo + "_xd": xd, algo + "_ud": ud, algo + "_p": p, algo + "_its": its, algo + "_calls": np.array(calls[0]),
                    algo + "_range": np.array([lo, hi])})
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_the_single_process_batch(pkg, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for algo, model in (("ptr", "rocket_landing"), ("scvx", "quadrotor")):
        pp = _pp(pkg, model, B_TOTAL)
        calls = [0]

        def count(n, calls=calls):
            calls[0] += 1
            return n
        xd, ud, p, its = _solve(pkg, algo, pp, count)
        assert list(r0[algo + "_range"]) == [0, 32] and list(r1[algo + "_range"]) == [32, 64]
        for nm, whole in (("xd", xd), ("ud", ud), ("p", p), ("its", its)):
            both = np.concatenate([r0[algo + "_" + nm], r1[algo + "_" + nm]], axis=0)
            assert np.array_equal(both, whole), (algo, nm)          # bit-identical: problems never interact
        # lockstep: every rank iterates until NO rank has an active problem -- as often as the single process did
        assert int(r0[algo + "_calls"]) == int(r1[algo + "_calls"]) == calls[0] == int(its.max())
        assert its.min() < (14 if algo == "ptr" else 12)                # the stopping criterion really ended problems early


def _group_solve(pkg, pp, all_reduce, pipelined):
    """the bench's loop: sub-batches on two streams, guesses on the device, one PTR iteration per window"""
    traj = pkg.TrajectoryProblem("rocket_landing")
    pars = pkg.PTR.Parameters(N=20, Nsub=8, iter_max=14, wvc=1e3, wtr=0.1, eps_abs=1e-4, eps_rel=1e-5, feas_tol=1e-3)
    grp = pkg.PTR.SCPProblemGroup(pars, traj, batch_capacity=pp.shape[0], streams=2)
    pkg.PTR.group_upload(grp, pp, device_guess=True)
    pkg.PTR.group_restart(grp)
    n_it = pkg.PTR.group_run_resident(grp, all_reduce, 1, pipelined=pipelined)
    if hasattr(all_reduce, "flush"):
        all_reduce.flush()
    pkg.PTR.group_sync(grp)
    sol, hist = pkg.PTR.group_collect(grp)
    grp.close()
    return sol.xd, sol.ud, sol.p, np.asarray(sol.iterations), n_it


def _group_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pp = _pp(pkg, "rocket_landing", B_TOTAL)
    lo, hi = pkg.dist.shard_range(B_TOTAL, rank, world)
    xd, ud, p, its, n_it = _group_solve(pkg, pp[lo:hi], pkg.dist.make_lagged_all_reduce(dist, device="cpu"), True)
    np.savez(os.path.join(out_dir, "grank%d.npz" % rank), xd=xd, ud=ud, p=p, its=its, n_it=np.array(n_it))
    dist.destroy_process_group()


def test_pipelined_group_loop_with_the_lagged_all_reduce_reproduces_the_single_process_batch(pkg, tmp_path):
    """bench.py's multi-GPU loop (group_run_resident(pipelined=True): window k + 1 enqueued before the count of window k is read with
    scp_ptr_poll_iteration; the all-reduce issued asynchronously and read one window later) on two ranks sharing the GPU: the
    union of the shards equals the plain single-process loop bit for bit, both ranks enqueue the same number of windows, and
    problems stop at their own iterations."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_group_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    r0, r1 = np.load(tmp_path / "grank0.npz"), np.load(tmp_path / "grank1.npz")
    pp = _pp(pkg, "rocket_landing", B_TOTAL)
    xd, ud, p, its, n_plain = _group_solve(pkg, pp, None, False)
    for nm, whole in (("xd", xd), ("ud", ud), ("p", p), ("its", its)):
        assert np.array_equal(np.concatenate([r0[nm], r1[nm]], axis=0), whole), nm
    assert int(r0["n_it"]) == int(r1["n_it"])                        # lockstep
    assert n_plain == int(its.max()) <= int(r0["n_it"]) <= 14 and its.min() < 14
    # the pipelined loop alone (one process, no collective) is the same computation too
    xd2, ud2, p2, its2, _ = _group_solve(pkg, pp, pkg.dist.make_lagged_all_reduce(None), True)
    assert np.array_equal(xd2, xd) and np.array_equal(its2, its)


def _sharded_solve(pkg, pp, comm, lookahead=1, streams=2):
    """the same loop BEHIND the C ABI: scp_ptr_run_sharded (include/scp_mi355x.h, "Multi-GPU")"""
    traj = pkg.TrajectoryProblem("rocket_landing")
    pars = pkg.PTR.Parameters(N=20, Nsub=8, iter_max=14, wvc=1e3, wtr=0.1, eps_abs=1e-4, eps_rel=1e-5, feas_tol=1e-3)
    grp = pkg.PTR.SCPProblemGroup(pars, traj, batch_capacity=pp.shape[0], streams=streams)
    pkg.PTR.group_upload(grp, pp, device_guess=True)
    pkg.PTR.group_restart(grp)
    n_it, n_coll = pkg.PTR.group_run_sharded(grp, comm, lookahead)
    pkg.PTR.group_sync(grp)
    sol, hist = pkg.PTR.group_collect(grp)
    grp.close()
    return sol.xd, sol.ud, sol.p, np.asarray(sol.iterations), n_it, n_coll


def test_sharded_loop_behind_the_c_abi_with_a_one_rank_rccl_communicator(pkg):
    """scp_ptr_run_sharded -- windows enqueued ahead, the device-resident active counts of the sub-batches summed by a kernel and
    all-reduced by RCCL (ncclAllReduce on a ONE-rank communicator created through scp_comm_unique_id / scp_comm_create: the same
    calls, kernels and stream ordering as on 8 GPUs; two ranks cannot share one GPU under RCCL) -- reproduces the plain
    single-process loop bit for bit, stops at the iteration the last problem stops, and issues one collective per window."""
    pp = _pp(pkg, "rocket_landing", B_TOTAL)
    xd, ud, p, its, n_plain = _group_solve(pkg, pp, None, False)
    comm = pkg.dist.Communicator(None, device=0, force_rccl=True)
    try:
        assert comm.all_reduce_sum(41) == 41
        xs, us, ps, its_s, n_it, n_coll = _sharded_solve(pkg, pp, comm)
    finally:
        comm.close()
    assert np.array_equal(xs, xd) and np.array_equal(us, ud) and np.array_equal(ps, p) and np.array_equal(its_s, its)
    assert n_it == n_plain == int(its.max()) and its.min() < 14
    assert n_it + 1 <= n_coll <= n_it + 2            # one collective per window; one window is enqueued past the last active one
    # without a communicator (comm = NULL): the same loop, no RCCL; and with two iterations per window
    x0, u0, p0, its0, n0, c0 = _sharded_solve(pkg, pp, None)
    assert np.array_equal(x0, xd) and np.array_equal(its0, its) and n0 == n_plain and c0 == 0
    x2, u2, p2, its2, n2, c2 = _sharded_solve(pkg, pp, None, lookahead=2)
    assert np.array_equal(x2, xd) and np.array_equal(its2, its) and n_plain <= n2 <= n_plain + 1


def test_sharded_loop_at_the_cap_of_sixteen_parts_and_beyond(pkg):
    """scp_ptr_run_sharded accepts at most 16 sub-batch handles.  AT the cap (16 parts of 4 problems, uneven
    iteration counts) the loop is the same computation as the plain one; ONE MORE part is refused with SCP_ERR_BAD_ARGUMENT before
    anything is enqueued."""
    pp = _pp(pkg, "rocket_landing", B_TOTAL)
    xd, ud, p, its, n_plain = _group_solve(pkg, pp, None, False)
    x16, u16, p16, its16, n16, c16 = _sharded_solve(pkg, pp, None, streams=16)
    assert np.array_equal(x16, xd) and np.array_equal(u16, ud) and np.array_equal(p16, p) and np.array_equal(its16, its) and n16 == n_plain
    traj = pkg.TrajectoryProblem("rocket_landing")
    pars = pkg.PTR.Parameters(N=20, Nsub=8, iter_max=14, wvc=1e3, wtr=0.1, eps_abs=1e-4, eps_rel=1e-5, feas_tol=1e-3)
    grp = pkg.PTR.SCPProblemGroup(pars, traj, batch_capacity=pp.shape[0], streams=17)
    try:
        pkg.PTR.group_upload(grp, pp, device_guess=True)
        pkg.PTR.group_restart(grp)
        with pytest.raises(pkg._lib.ScpError) as e:
            pkg.PTR.group_run_sharded(grp, None, 1)
        assert e.value.code == 1, str(e.value)        # SCP_ERR_BAD_ARGUMENT
        pkg.PTR.group_sync(grp)
        sol, _ = pkg.PTR.group_collect(grp)
        assert (np.asarray(sol.iterations) == 0).all()        # nothing ran
    finally:
        grp.close()
