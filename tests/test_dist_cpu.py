"""N > 1 path on CPU: world_size-2 gloo processes run the sharded PTR driver loop (scptoolbox.jl_amd/dist.py)
with a stand-in for the device iteration, checking that the shards tile the batch, that the per-iteration
all-reduce keeps the ranks in lockstep and that the loop stops only when NO rank has active problems."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_tiles_the_batch(pkg):
    for n, w in ((1024, 8), (1000, 3), (5, 8), (4096, 2)):
        cover = []
        for r in range(w):
            lo, hi = pkg.dist.shard_range(n, r, w)
            cover += list(range(lo, hi))
        assert cover == list(range(n))
        sizes = [pkg.dist.shard_range(n, r, w)[1] - pkg.dist.shard_range(n, r, w)[0] for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def test_strong_and_weak_scaling_coincide_on_one_gpu(pkg):
    """`bench.py --scaling strong --global-batch 4096` on ONE GPU is the default (weak) workload: same batch, same instances;
    on 8 GPUs the strong shards tile the 4096 instances the weak run's rank 0 holds alone."""
    import bench
    wb = bench.WORKLOADS["rocket_landing"][4]
    assert wb == 4096
    assert bench.local_shard(pkg, "strong", wb, 0, 4096, 0, 1) == bench.local_shard(pkg, "weak", wb, 0, 4096, 0, 1) == (4096, 0)
    shards = [bench.local_shard(pkg, "strong", wb, 0, 4096, r, 8) for r in range(8)]
    assert [s[0] for s in shards] == [512] * 8 and [s[1] for s in shards] == [512 * r for r in range(8)]
    assert [bench.local_shard(pkg, "weak", wb, 0, 4096, r, 8) for r in range(8)] == [(4096, 4096 * r) for r in range(8)]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = pkg.dist.shard_range(10, rank, world)
    # stand-in for scp_ptr_iterate: problem i stops after (3 + i) iterations
    stop_at = np.arange(lo, hi) + 3
    state = {"it": 0}

    def iterate():
        state["it"] += 1
        return int((stop_at > state["it"]).sum())
    ar = pkg.dist.make_all_reduce(dist)
    n_calls = pkg.dist.run_sharded(iterate, ar)
    # the non-blocking variant: the count of window k is read while window k + 1 is being enqueued -> one extra window, same on
    # every rank
    state["it"] = 0
    lag = pkg.dist.make_lagged_all_reduce(dist)
    n_lag = pkg.dist.run_sharded(iterate, lag)
    assert lag.flush() == 0
    gathered = pkg.dist.gather_concat([stop_at.astype(np.float64)], dist)
    q.put((rank, n_calls, (lo, hi), gathered[0].tolist(), n_lag))
    dist.destroy_process_group()


def test_sharded_loop_lockstep_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the slowest problem (index 9) stops after 12 iterations: BOTH ranks must have iterated 12 times
    assert [r[1] for r in res] == [12, 12]
    assert [r[4] for r in res] == [13, 13]          # lagged all-reduce: one more (no-op) window, in lockstep
    assert res[0][2] == (0, 5) and res[1][2] == (5, 10)
    assert res[0][3] == list(np.arange(10) + 3.0) == res[1][3]


class _FakeLib:
    """stand-in for libscp_mi355x's PTR iteration entry points: handle = dict(stop_at, iter_max, it); a problem is active after
    iteration k iff k < its stop iteration and k < iter_max; nothing is enqueued beyond iter_max (as scp_ptr_iterate_async)"""

    def __init__(self):
        self.log = []

    @staticmethod
    def _count(h, k):
        return 0 if k >= h["iter_max"] else int((h["stop_at"] > k).sum())

    def scp_ptr_iterate_async(self, h):
        h["it"] += 1
        self.log.append(("enq", id(h), h["it"]))
        return 0

    def scp_ptr_poll(self, h, ref):
        ref._obj.value = self._count(h, h["it"])
        return 0

    def scp_ptr_poll_iteration(self, h, k, ref):
        assert 1 <= k <= h["it"], "only an iteration that has been enqueued can be polled"
        self.log.append(("poll", id(h), k, h["it"]))
        ref._obj.value = self._count(h, k)
        return 0


class _FakePart:
    def __init__(self, stop_at, iter_max):
        self.handle = dict(stop_at=np.asarray(stop_at), iter_max=iter_max, it=0)


class _FakeGroup:
    def __init__(self, stop_at, iter_max, streams=2):
        cut = np.array_split(np.asarray(stop_at), streams)
        self.parts = [_FakePart(c, iter_max) for c in cut]
        self.pars = type("P", (), {"iter_max": iter_max})()


def _pipelined_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fake = _FakeLib()
    pkg._lib.lib = lambda: fake
    pkg._lib.check = lambda rc, h=None: None
    out = {}
    for name, stops, iter_max in (("early", np.arange(10) + 3, 20), ("fixed", np.full(10, 10 ** 6), 15)):
        lo, hi = pkg.dist.shard_range(10, rank, world)
        grp = _FakeGroup(stops[lo:hi], iter_max)
        lag = pkg.dist.make_lagged_all_reduce(dist)
        fake.log.clear()
        n_it = pkg.PTR.group_run_resident(grp, lag, 1, pipelined=True)
        lag.flush()
        enq = max(p.handle["it"] for p in grp.parts)
        # every poll of window k happened AFTER window k + 1 was enqueued
        ahead = all(ev[3] == ev[2] + 1 for ev in fake.log if ev[0] == "poll")
        out[name] = (n_it, enq, ahead)
    q.put((rank, out))
    dist.destroy_process_group()


def test_pipelined_group_loop_lockstep_gloo():
    """group_run_resident(pipelined=True) (the multi-GPU loop of bench.py) over gloo with a stand-in for the library: the count of
    window k is polled with window k + 1 already enqueued, both ranks enqueue the same number of windows -- the slowest problem
    anywhere (12 iterations) + one window for the pipeline + one for the lagged collective --, and a fixed-iteration run stops at
    iter_max with nothing counted beyond it."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        n_it, enq, ahead = res[rank]["early"]
        assert ahead and enq == 12 + 2 and n_it == 12            # problem 9 (rank 1) stops after 12 iterations: both ranks follow;
                                                                 # the two no-op windows behind it are enqueued but not counted
        n_it, enq, ahead = res[rank]["fixed"]
        assert ahead and n_it == 15 and enq == 15 + 2            # iter_max = 15: two more (empty) calls, none counted


def test_sharded_loop_lockstep_gloo_eight_ranks_uneven_shards():
    """the 8-GPU shape on CPU: 8 gloo ranks over 10 problems (shards of 2, 2, 1, 1, 1, 1, 1, 1 -- scp_shard_range / dist.shard_range),
    the slowest problem on the LAST rank: every rank iterates 12 times (13 windows with the lagged collective), the shards tile the
    batch in rank order."""
    world = 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [12] * world and [r[4] for r in res] == [13] * world
    assert [r[2][1] - r[2][0] for r in res] == [2, 2, 1, 1, 1, 1, 1, 1]
    assert [r[2][0] for r in res] == [0, 2, 4, 5, 6, 7, 8, 9]
    assert all(r[3] == list(np.arange(10) + 3.0) for r in res)


def _shipped_loop_worker(rank, world, port, q):
    """two gloo ranks drive THE SHIPPED window loop (csrc/sharded_loop.hpp = the body of scp_ptr_run_sharded, compiled for the host:
    oracle/sharded_host.cpp) -- enqueue(w) advances a stand-in for the PTR iterations of the window and issues the all-reduce of the rank's
    active count asynchronously (as comm_reduce_window does on the RCCL stream), wait(w) finishes it (hipEventSynchronize in the library)"""
    import ctypes
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    import subprocess
    if rank == 0:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libsharded_host.so"))
    ENQ = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int)
    WAIT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong))
    ABORT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong)
    out = {}
    for name, stops, iter_max, lookahead in (("early", np.arange(10) + 3, 20, 1), ("fixed", np.full(10, 10 ** 6), 15, 1), ("windows_of_4", np.arange(10) + 3, 20, 4),
                                              ("uneven", np.array([2, 2, 2, 2, 2, 2, 2, 2, 2, 17]), 20, 1),
                                              ("rank_1_fails_in_window_3", np.full(10, 10 ** 6), 15, 1), ("rank_0_fails_in_window_0", np.full(10, 10 ** 6), 15, 1)):
        fail_rank, fail_window = (1, 3) if name.startswith("rank_1") else ((0, 0) if name.startswith("rank_0") else (-1, -1))
        lo, hi = pkg.dist.shard_range(10, rank, world)
        stop_at = stops[lo:hi]
        st = dict(it=0, log=[], pending={}, enq=0)

        def enqueue(user, w, st=st, stop_at=stop_at, iter_max=iter_max, lookahead=lookahead, fail_rank=fail_rank, fail_window=fail_window):
            if rank == fail_rank and w == fail_window:        # a local failure BEFORE this window's collective is issued (e.g. a HIP error)
                st["log"].append(("fail", w))
                return 5
            for _ in range(lookahead):
                if st["it"] < iter_max:             # nothing is enqueued beyond iter_max (scp_ptr_iterate_async)
                    st["it"] += 1
            st["enq"] += 1
            n_local = 0 if st["it"] >= iter_max and w * lookahead >= iter_max else int((stop_at > st["it"]).sum())
            t = torch.tensor([n_local], dtype=torch.int64)
            st["pending"][w] = (t, dist.all_reduce(t, async_op=True))
            st["log"].append(("enq", w))
            return 0

        def wait(user, w, out_n, st=st):
            t, work = st["pending"].pop(w)
            work.wait()
            out_n[0] = int(t.item())
            st["log"].append(("wait", w, st["enq"]))
            return 0
        def abort(user, w, sentinel, st=st):
            # what comm_reduce_window does with the sentinel in the send slot: the window's collective IS issued, carrying the failure
            t = torch.tensor([int(sentinel)], dtype=torch.int64)
            st["pending"][w] = (t, dist.all_reduce(t, async_op=True))
            st["log"].append(("abort", w))
            return 0
        windows = L.sharded_host_windows(iter_max, lookahead)
        done = ctypes.c_int(-1)
        rc = L.sharded_host_loop(windows, ENQ(enqueue), WAIT(wait), ABORT(abort), None, ctypes.byref(done))
        if rc == L.sharded_host_peer_failed():
            rc = "peer"
        for t, work in st["pending"].values():      # the collective of the window enqueued ahead (every rank issued it)
            work.wait()
        ahead = all(ev[2] >= ev[1] + 2 or ev[1] + 1 >= windows for ev in st["log"] if ev[0] == "wait")   # window w + 1 enqueued before the count of w is read
        out[name] = (rc, L.sharded_host_iterations(0, done.value, lookahead, iter_max), st["enq"], bool(ahead), windows)
        if fail_rank >= 0:
            out[name] = (rc, [ev for ev in st["log"] if ev[0] in ("fail", "abort")], sorted(w for ev in st["log"] if ev[0] == "enq" for w in [ev[1]]))
    q.put((rank, out))
    dist.destroy_process_group()


def test_shipped_sharded_window_loop_with_two_gloo_ranks():
    """the pipelining logic that SHIPS (csrc/sharded_loop.hpp, included by scp_api.hip::scp_ptr_run_sharded) is the logic
    tested with more than one rank: both ranks enqueue the same number of windows, stop together after the slowest problem anywhere, a count is
    read only with the next window already enqueued, a fixed-iteration run (eps = 0) stops at iter_max, uneven shards (all of one rank's
    problems stop early) keep the collectives matched."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shipped_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for name in ("early", "fixed", "windows_of_4", "uneven"):
        assert res[0][name] == res[1][name], (name, res)
    rc, n_it, enq, ahead, windows = res[0]["early"]        # problem 9 (rank 1) is active for 12 iterations: the first window with a zero count is the 12th
    assert rc == 0 and ahead and n_it == 12 and enq == 13 and windows == 22
    rc, n_it, enq, ahead, windows = res[0]["fixed"]
    assert rc == 0 and ahead and n_it == 15 and enq == 17 and windows == 17      # all 15 + 2 windows (the count of iteration 15 is still > 0 with eps = 0; nothing runs in the last two)
    rc, n_it, enq, ahead, windows = res[0]["windows_of_4"]  # 12 iterations = 3 windows of 4
    assert rc == 0 and ahead and n_it == 12 and enq == 4 and windows == 7
    rc, n_it, enq, ahead, windows = res[0]["uneven"]        # rank 0's problems stop after 2 iterations, rank 1 holds one that runs 17
    assert rc == 0 and ahead and n_it == 17 and enq == 18
    # a rank that FAILS while enqueuing a window: it reports its own error after contributing the failure sentinel to that
    # window's and the next window's collective; the other rank reads a negative count for that window and leaves with "a peer failed";
    # every collective either rank started is matched (the workers wait for all of them: the test would hang otherwise)
    rc1, ev1, enq1 = res[1]["rank_1_fails_in_window_3"]
    rc0, ev0, enq0 = res[0]["rank_1_fails_in_window_3"]
    assert rc1 == 5 and ev1 == [("fail", 3), ("abort", 3), ("abort", 4)] and enq1 == [0, 1, 2]
    assert rc0 == "peer" and ev0 == [] and enq0 == [0, 1, 2, 3, 4]
    rc0, ev0, enq0 = res[0]["rank_0_fails_in_window_0"]
    rc1, ev1, enq1 = res[1]["rank_0_fails_in_window_0"]
    assert rc0 == 5 and ev0 == [("fail", 0), ("abort", 0), ("abort", 1)] and enq0 == []
    assert rc1 == "peer" and ev1 == [] and enq1 == [0, 1]


def _preflight_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 1 asks for a device that does not exist anywhere; rank 0 for device 0 (absent on this box as well, present on a GPU box):
        # either way at least one rank fails LOCALLY, before the collective part of the create
        pkg.dist.Communicator(dist, device=0 if rank == 0 else 4096)
        out = "created"
    except pkg._lib.ScpError as e:
        out = str(e)
    q.put((rank, out))
    dist.destroy_process_group()


def test_communicator_preflight_is_agreed_before_the_collective_create():
    """a rank whose LOCAL communicator set-up fails (RCCL missing, no such device, stream creation) must not leave the other
    ranks inside ncclCommInitRank.  `scp_comm_preflight` runs those local steps alone, `dist.Communicator` gathers the results over the
    host's process group and EVERY rank raises -- none hangs, none reaches scp_comm_create."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_preflight_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "preflight failed on rank(s)" in res[0] and "preflight failed on rank(s)" in res[1], res
    assert res[0] == res[1]          # the same ranks and the same first error everywhere
