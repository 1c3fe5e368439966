// Shared definitions of the batched structured interior-point solver K3: kernel arguments, exit codes, wavefront
// reductions.  The solver itself is csrc/ipm2_kernel.hpp (+ ipm2_newton.hpp, ipm2_run.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage_problem.hpp"

namespace scp {

struct IpmArgs {
    int B, N;
    int max_iter, nref, stall, split_step;
    double feastol, abstol, reltol, reg, ref_gap, ref_tol;
    const double* slab;  // problem data [B][slab_stride]
    long slab_stride;
    double* work;  // workspace [B][work_stride]
    long work_stride;
    // outputs
    double* z_out;    // [B][N*nz] scaled primal
    double* p_out;    // [B][npa]
    int* status;      // [B]: 0 OPTIMAL, 1 ALMOST_OPTIMAL, 2 ITERATION_LIMIT, 3 NUMERICAL_ERROR
    int* iters;       // [B]
    double* info;     // [B][8]: pcost(+const), dcost, gap, pres, dres, relgap, merit, best_it
    const int* active;  // optional [B]: problems with active[b] == 0 are skipped
    // warm start (see scp_ptr_params.ipm_warm): allowed by the host only inside a running PTR loop (iteration >= 2).  The
    // solver keeps NWL SNAPSHOTS of every solve -- the iterates at which the complementarity measure mu = gap / degree first
    // fell below warm_mu[l] (well-centred interior points; l = 0 coarse ... NWL - 1 very fine) -- and the next solve of that problem
    // starts from the finest level l with prev_dev <= warm_dev[l] whose snapshot exists (warm_dev[0] = infinity; level 0 only where
    // cold solves are slow, warm_min_cold).  Round 6: four levels (two before).
    static constexpr int NWL = 4;
    int warm_allowed, warm_min_cold;
    double warm_mu[NWL], warm_dev[NWL];
    const double* prev_dev;   // [B] deviation of the previous solution from its reference
    int* cold_iters;          // [B] iterations of the last cold solve of each problem (in/out)
    int* snap;                // [B] bit l: the level-l snapshot in the workspace is valid (in/out)
    long long* prof;    // optional [B][8] phase counters (100 MHz wall clock ticks)
};

enum { IPM_OPTIMAL = 0, IPM_ALMOST = 1, IPM_ITERLIM = 2, IPM_NUMERR = 3 };

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

}  // namespace scp
