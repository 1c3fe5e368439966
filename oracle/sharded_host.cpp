// TEST INFRASTRUCTURE (not product code): host build of the product's sharded-loop control flow (csrc/sharded_loop.hpp, the window loop of
// scp_ptr_run_sharded, scp_api.hip) behind C callbacks, so that tests/test_dist_cpu.py can drive THE SHIPPED logic with two gloo ranks on the CPU
// -- enqueue = advance a stand-in for the PTR iteration + issue the (asynchronous) all-reduce, wait = finish it.
#include "../scptoolbox.jl_amd/csrc/sharded_loop.hpp"

extern "C" {
typedef int (*sharded_enqueue_cb)(void* user, int window);
typedef int (*sharded_wait_cb)(void* user, int window, long long* global_count);

int sharded_host_windows(int iter_max, int lookahead) { return scp::sharded_windows(iter_max, lookahead); }
int sharded_host_iterations(int it0, int done_window, int lookahead, int iter_max) { return scp::sharded_iterations(it0, done_window, lookahead, iter_max); }
typedef int (*sharded_abort_cb)(void* user, int window, long long sentinel);
int sharded_host_loop(int windows, sharded_enqueue_cb enqueue, sharded_wait_cb wait, sharded_abort_cb abort, void* user, int* done_window)
{
    return scp::sharded_window_loop(windows, [&](int w) { return enqueue(user, w); }, [&](int w, long long* n) { return wait(user, w, n); },
                                    [&](int w) { return abort ? abort(user, w, scp::SHARDED_SENTINEL) : 0; }, done_window);
}
int sharded_host_peer_failed(void) { return scp::SHARDED_PEER_FAILED; }
}
