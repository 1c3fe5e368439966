// Host-side control flow of the sharded PTR loop (scp_ptr_run_sharded), free of HIP so that the SAME code runs in
// the library and -- compiled for the host with stand-in callbacks -- in the multi-process CPU tests (oracle/sharded_host.cpp,
// tests/test_dist_cpu.py: gloo, world size 2).
//
// A run is cut into WINDOWS of `lookahead` PTR iterations.  `enqueue(w)` puts window w on the device streams together with the (non-blocking)
// all-reduce of the number of problems still active after it; `wait(w, &n)` blocks until that global count has arrived.  Window w + 1 is
// enqueued BEFORE the count of window w is read, so no stream drains at a window boundary; every rank enqueues the same windows in the same
// order, so the collectives match.  The loop ends with the first window whose global count is zero (nothing was active anywhere) or with the
// last window.
#pragma once
#include <algorithm>

namespace scp {

// windows a run of iter_max iterations is cut into: the iterations, plus one window for the pipeline and one whose count must read 0
inline int sharded_windows(int iter_max, int lookahead) { return (iter_max + lookahead - 1) / lookahead + 2; }

// iterations executed when the loop ended with window `done_window` of a run that started at iteration it0
inline int sharded_iterations(int it0, int done_window, int lookahead, int iter_max) { return std::min(it0 + (done_window + 1) * lookahead, iter_max); }

// A rank that fails while enqueuing window k must not simply return: its peers have issued (or are about to issue) the collectives of windows
// k and k + 1 and would wait for its contribution for ever.  It contributes a large negative SENTINEL to exactly those two collectives
// (`abort(w)`) and returns its error; a peer that reads a negative count for window k has, by the order of the loop, issued the collectives up
// to k + 1 and nothing beyond -- every collective that was started is matched -- and returns SHARDED_PEER_FAILED.  
constexpr long long SHARDED_SENTINEL = -(1LL << 40);      // sums of active counts (< 2^31 per rank) cannot cancel it
constexpr int SHARDED_PEER_FAILED = -1000;                // mapped to SCP_ERR_PEER by the library

template <class Enqueue, class Wait, class Abort>
int sharded_window_loop(int windows, Enqueue&& enqueue, Wait&& wait, Abort&& abort, int* done_window)
{
    auto fail = [&](int k, int rc) { abort(k); if (k + 1 < windows) abort(k + 1); return rc; };
    int rc = enqueue(0);
    if (rc) return fail(0, rc);
    int w = 0;
    while (true) {
        if (w + 1 < windows) { rc = enqueue(w + 1); if (rc) return fail(w + 1, rc); }      // window w + 1 is on the device BEFORE the count of window w is read
        long long n = 0;
        rc = wait(w, &n);
        if (rc) return rc;
        if (n < 0) { *done_window = w; return SHARDED_PEER_FAILED; }
        if (n == 0) { *done_window = w; break; }
        w++;
        if (w >= windows) { *done_window = windows - 1; break; }
    }
    return 0;
}

}  // namespace scp
