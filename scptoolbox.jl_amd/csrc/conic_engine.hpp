// Device engine of the generic batched conic solver: owns the uploaded schedule (conic_symbolic.hpp) and the
// interleaved per-problem buffers, launches conic_ipm_kernel (conic_api.hip).  Used by the C ABI of
// include/scp_conic.h and, device-resident, by the generic SCP path (scp_generic.hip: SCvx / GuSTO / PTR q_tr != Inf).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <vector>

#include "conic_ipm.hpp"
#include "conic_symbolic.hpp"

namespace scp {
namespace conic {

struct Engine {
    Symbolic sym;
    Sched sched{};           // device pointers
    // Ordering (conic_symbolic.hpp): the primary schedule is the NESTED-DISSECTION one when the program has the chain
    // structure (5-10x fewer elimination levels = workgroup barriers).  Its pivots are less protected than those of the
    // sequential order on degenerate LPs (Starship), so a second, sequential schedule is kept: every problem the primary
    // pass does not bring to OPTIMAL (or to an infeasibility certificate) is re-solved with it in a second launch, and the
    // engine switches to the sequential schedule for good when that happens to more than a quarter of a batch.
    // SCP_CONIC_ORDER = seq | nd | auto (default auto).
    Symbolic sym_fb;
    Sched sched_fb{};
    bool has_fb = false;
    int *fb_mask = nullptr, *fb_count = nullptr;
    long long n_fallback = 0, n_launched = 0, n_rescued = 0;   // problems re-solved by the fallback pass / solved / rescued by it
    int cap = 0, BS = 0;     // batch capacity, interleave stride (cap rounded up to 64)
    int device = 0;
    // worker waves per group of 64 problems (1..16): tuning aid SCP_CONIC_WAVES, default 16 (a full 1024-thread workgroup)
    int waves_per_group = std::getenv("SCP_CONIC_WAVES") ? std::atoi(std::getenv("SCP_CONIC_WAVES")) : 16;
    // sub-workers per wave (1, 4 or 16 -> 64, 16 or 4 problems per wave); 0 = chosen from the batch size (tuning aid
    // SCP_CONIC_SUB)
    int sub_workers = std::getenv("SCP_CONIC_SUB") ? std::atoi(std::getenv("SCP_CONIC_SUB")) : 0;
    std::vector<void*> allocs;
    // interleaved inputs [len][BS] and shared copies [len]
    double *c = nullptr, *b = nullptr, *h = nullptr, *Gx = nullptr, *Ax = nullptr, *Px = nullptr;
    double *c_sh = nullptr, *b_sh = nullptr, *h_sh = nullptr, *Gx_sh = nullptr, *Ax_sh = nullptr, *Px_sh = nullptr;
    // interleaved solution [len][BS]
    double *x = nullptr, *y = nullptr, *z = nullptr, *s = nullptr;
    double* work = nullptr;  // everything else
    int *status = nullptr, *iters = nullptr;
    double* info = nullptr;  // [8][BS]
    long long bytes_per_problem = 0;
    std::string err;

    // analyse + upload + allocate; returns scp_status
    int create(int n, int p, int m, int l, const std::vector<int>& q, const Csc& P, const Csc& A, const Csc& G,
               const int* perm, int capacity, int dev);
    void destroy();
    // enqueue the solve of problems [0, B) on `stream`; shared_mask as in scp_conic.h; active: optional int[B]
    // (problems with active[t] == 0 are skipped)
    int launch(hipStream_t stream, int B, const Opts& o, unsigned shared_mask, const int* active = nullptr);
};

// [len, B] column-major (batch last) <-> interleaved [len][BS]; device pointers
int transpose_to_interleaved(hipStream_t st, const double* src, double* dst, long len, int B, int BS);
int transpose_from_interleaved(hipStream_t st, const double* src, double* dst, long len, int B, int BS);

}  // namespace conic
}  // namespace scp
