"""Batch sharding of independent SCP instances across GPUs (one process per GPU).

The reference has no parallelism at all; the Monte-Carlo batch is the
data-parallel axis: problems are independent, so the batch is split into contiguous ranges
with NO data-path collective.  The only collective is the per-iteration all-reduce of the
number of still-active problems (8 bytes), which keeps the ranks' PTR loops in lockstep for
the global stopping decision -- RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous [lo, hi) of the global batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def make_all_reduce(dist=None, device="cpu"):
    """Returns f(n_local) -> n_global (SUM).  `dist` = torch.distributed (initialised) or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return lambda n: int(n)
    import torch
    buf = torch.zeros(1, dtype=torch.int64, device=device)

    def f(n):
        buf[0] = int(n)
        dist.all_reduce(buf)
        return int(buf.item())
    return f


def make_lagged_all_reduce(dist=None, device="cpu"):
    """Non-blocking variant: f(n_local) ENQUEUES the all-reduce of this window's count (async_op) and returns the global count of
    the PREVIOUS window (a large number on the first call), so the host never waits for a collective that was just issued -- by
    the time a result is read the next window of iterations has been enqueued behind it.  Every rank sees the same lagged
    sequence, so the ranks stay in lockstep; the loop runs one extra window after global convergence (problems that have stopped
    are skipped on the device: those launches are no-ops).  `f.flush()` returns the last window's count (blocking)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        state = {"prev": None}

        def f1(n):
            prev, state["prev"] = state["prev"], int(n)
            return 1 << 62 if prev is None else prev
        def flush1():
            prev, state["prev"] = state["prev"], None
            return 0 if prev is None else prev
        f1.flush = flush1
        return f1
    import torch
    bufs = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(2)]
    state = {"work": None, "slot": 0}

    def wait_prev():
        w, state["work"] = state["work"], None
        if w is None:
            return None
        w.wait()
        return int(bufs[1 - state["slot"]].item())      # the buffer of the previous call

    def f(n):
        prev = wait_prev()
        b = bufs[state["slot"]]
        b[0] = int(n)
        state["work"] = dist.all_reduce(b, async_op=True)
        state["slot"] = 1 - state["slot"]
        return 1 << 62 if prev is None else prev

    def flush():
        v = wait_prev()
        return 0 if v is None else v
    f.flush = flush
    return f


class Communicator:
    """The library's own communicator (include/scp_mi355x.h, "Multi-GPU": scp_comm_*): RCCL inside libscp_mi355x.so, the
    all-reduce of the device-resident active count enqueued by `scp_ptr_run_sharded` -- what a Julia host binds with `ccall`
    (INTEGRATION.md).  `dist` (torch.distributed, initialised, any backend) only ships rank 0's 128-byte id to the other ranks, as
    MPI.bcast would; dist = None / world 1: the single-process communicator (no RCCL)."""

    def __init__(self, dist=None, device=0, force_rccl=False):
        import ctypes
        from . import _lib
        L = _lib.lib()
        world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        self.rank, self.world, self.device = rank, world, device
        self._h = ctypes.c_void_p()
        idb = None
        if world > 1:
            # every rank's LOCAL steps first (load RCCL, select the device, create a stream), and agreement on them BEFORE the collective
            # scp_comm_create: a rank that failed locally would leave the others inside ncclCommInitRank for ever
            rc = int(L.scp_comm_preflight(int(device)))
            mine = (rank, rc, L.scp_comm_last_error(None).decode(errors="replace") if rc else "")
            allr = [None] * world
            dist.all_gather_object(allr, mine)
            bad = [r for r in allr if r[1] != 0]
            if bad:
                raise _lib.ScpError(bad[0][1], "communicator preflight failed on rank(s) %s: %s" % ([r[0] for r in bad], bad[0][2]))
        if world > 1 or force_rccl:        # force_rccl: a ONE-rank RCCL communicator (tests the RCCL path on a single GPU)
            buf = ctypes.create_string_buffer(128)
            box = [None]
            if rank == 0:
                rc = L.scp_comm_unique_id(buf)
                # (a failure on rank 0 is SHIPPED, not raised before the broadcast: the other ranks would wait in it for ever)
                box = [bytes(buf.raw)] if rc == 0 else [(int(rc), L.scp_comm_last_error(None).decode(errors="replace"))]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            if not isinstance(box[0], bytes):
                raise _lib.ScpError(*box[0])
            idb = ctypes.create_string_buffer(box[0], 128)
        rc = L.scp_comm_create(idb, rank, world, device, ctypes.byref(self._h))
        if rc != 0:
            self._h = ctypes.c_void_p()
            raise _lib.ScpError(rc, L.scp_comm_last_error(None).decode(errors="replace"))

    def all_reduce_sum(self, n):
        import ctypes
        from . import _lib
        v = ctypes.c_longlong(int(n))
        rc = _lib.lib().scp_comm_all_reduce_sum_i64(self._h, ctypes.byref(v))
        if rc != 0:
            raise _lib.ScpError(rc, _lib.lib().scp_comm_last_error(self._h).decode(errors="replace"))
        return int(v.value)

    def close(self):
        if self._h:
            from . import _lib
            _lib.lib().scp_comm_destroy(self._h)
            import ctypes
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_sharded(iterate, all_reduce, max_calls=10 ** 6):
    """Drive `iterate() -> n_active_local` until no problem is active on ANY rank.
    Every rank calls `iterate` the same number of times (ranks whose problems have all stopped keep
    calling it -- it is then a no-op on the device -- so that the collective stays matched)."""
    n = 0
    while n < max_calls:
        n_local = iterate()
        n += 1
        if all_reduce(n_local) <= 0:
            break
    return n


def gather_concat(arrays, dist=None):
    """Final host-side gather of per-rank result arrays (rank order == batch order)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return arrays
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, arrays)
    return [np.concatenate([o[i] for o in out], axis=0) for i in range(len(arrays))]
