// C ABI of the generic batched conic solver (include/scp_conic.h) + its device engine.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/scp_conic.h"
#include "../../include/scp_mi355x.h"
#include "conic_engine.hpp"

using namespace scp::conic;

namespace scp {
namespace conic {

// per-array base pointers of the interleaved layout: element e of problem t at  ptr[e * es + t * ts]
struct Arr { double* p; long es; long ts; };
struct ProbBase {
    Arr c, b, h, Gx, Ax, Px, x, y, z, s, Gt, Lx, Ux, Dinv, rhs, sol, res, cor, tmp, part, lam, wsc, ds, dz, corr, rz, eta, rx, ry;
};
__host__ __device__ inline BV bv(const Arr& a, long t) { return BV{a.p + t * a.ts, a.es}; }
__host__ __device__ inline CBV cbv(const Arr& a, long t) { return CBV{a.p + t * a.ts, a.es}; }

// Device execution context of the solver body (conic_ipm.hpp).  A workgroup of NW wavefronts owns PPW = 64 / SUB
// problems: lane -> (problem = lane % PPW, sub-worker = lane / PPW), so a wave works on SUB items at once for PPW problems
// each and the group has NW * SUB workers (SUB = 64: one problem per workgroup, 1024 workers).  SUB = 1: a wave's load of one element is a full 512-byte line (chip-filling
// batches); SUB = 4: 128-byte segments, but 4x the workgroups for the same batch -- the configuration for the north
// star's 4096-problem batch, which with SUB = 1 occupies only 64 of the 256 CUs.
template <int SUB>
struct DevCtx {
    static constexpr int PPW = 64 / SUB;
    // cooperative item groups (conic_ipm.hpp, Solver::pfor_coop): only when a workgroup owns ONE problem -- its workers are then
    // consecutive lanes of the waves, a group of G <= 64 of them lies inside one wave and sums with shuffles
    static constexpr bool COOP = SUB == 64;
    static constexpr bool COOP_EMU = false;
    __device__ __forceinline__ int coop_workers() const { return nworkers; }
    __device__ __forceinline__ double gsum(double v, int G) const
    {
        for (int msk = G >> 1; msk > 0; msk >>= 1) v += __shfl_xor(v, msk);
        return v;
    }
    double* red;   // LDS: SUB = 64: one partial per wave; otherwise [NW * SUB][PPW], one per worker
    int w, nworkers, prob;
    __device__ __forceinline__ int wid() const { return w; }
    __device__ __forceinline__ int nw() const { return nworkers; }
    __device__ __forceinline__ void barrier() const { __syncthreads(); }
    // Reductions over the 1 024 workers of a problem that owns its workgroup, two stages (round 5): a butterfly of shuffles inside the
    // wave, then one partial per wave through LDS, summed by every worker in wave order -- identical bits in every worker.  Rounds 2-4 had every worker add up ALL nworkers partials from LDS: 1 024 x 1 024 reads per
    // call when a workgroup owns one problem -- 1.0 ms per call, ~40 calls per IPM iteration (step lengths, residual norms, the
    // refinement test): more than the factorisation (profiles/r05_k5_phase_profile.txt).
    // (the chip-filling geometries, SUB < 64, keep the summation order of rounds 2-4 -- every worker adds the nworkers <= 256 partials
    //  in worker order -- so that large batches reproduce their earlier results bit for bit)
    __device__ __forceinline__ double sum(double v) const
    {
        if constexpr (SUB == 64) {
#pragma unroll
            for (int msk = 32; msk >= 1; msk >>= 1) v += __shfl_xor(v, msk);
            const int wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
            if ((threadIdx.x & 63) == 0) red[wave] = v;
            __syncthreads();
            double acc = 0.0;
            for (int i = 0; i < nwv; i++) acc += red[i];
            __syncthreads();
            return acc;
        } else {
            red[w * PPW + prob] = v;
            __syncthreads();
            double acc = 0.0;
            for (int i = 0; i < nworkers; i++) acc += red[i * PPW + prob];   // fixed order: identical in every worker
            __syncthreads();
            return acc;
        }
    }
    __device__ __forceinline__ double min(double v) const
    {
        if constexpr (SUB == 64) {
#pragma unroll
            for (int msk = 32; msk >= 1; msk >>= 1) v = fmin(v, __shfl_xor(v, msk));
            const int wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
            if ((threadIdx.x & 63) == 0) red[wave] = v;
            __syncthreads();
            double acc = red[0];
            for (int i = 1; i < nwv; i++) acc = fmin(acc, red[i]);
            __syncthreads();
            return acc;
        } else {
            red[w * PPW + prob] = v;
            __syncthreads();
            double acc = red[prob];
            for (int i = 1; i < nworkers; i++) acc = fmin(acc, red[i * PPW + prob]);
            __syncthreads();
            return acc;
        }
    }
    __device__ __forceinline__ bool any(bool v) const { return __syncthreads_or(v ? 1 : 0) != 0; }
};

constexpr int CONIC_MAX_WAVES = 16;

// Workgroup = (64 / SUB) problems x NW worker waves (blockDim.x = 64 NW); see conic_ipm.hpp.
// Two register budgets: MAXW = 16 (1024 threads, 128 VGPRs) and MAXW = 8 (512 threads, 256 VGPRs: no spills).
template <int MAXW, int SUB>
__global__ __launch_bounds__(64 * MAXW) void conic_ipm_kernel(Sched S, ProbBase PB, Opts O, int B, const int* active,
                                                             int* status, int* iters, double* info, long info_es)
{
    using Ctx = DevCtx<SUB>;
    constexpr int PPW = Ctx::PPW;
    __shared__ double red[MAXW * 64];
    Ctx cx;
    const int lane = threadIdx.x & 63;
    cx.red = red; cx.prob = lane % PPW; cx.w = (threadIdx.x >> 6) * SUB + lane / PPW; cx.nworkers = (blockDim.x >> 6) * SUB;
    const int t = blockIdx.x * PPW + cx.prob;     // < BS: padding lanes own (unused) storage of their own
    const bool live = t < B && (active == nullptr || active[t] != 0);
    Prob Q;
    Q.c = cbv(PB.c, t); Q.b = cbv(PB.b, t); Q.h = cbv(PB.h, t); Q.Gx = cbv(PB.Gx, t); Q.Ax = cbv(PB.Ax, t); Q.Px = cbv(PB.Px, t);
    Q.x = bv(PB.x, t); Q.y = bv(PB.y, t); Q.z = bv(PB.z, t); Q.s = bv(PB.s, t);
    Q.Gt = bv(PB.Gt, t); Q.Lx = bv(PB.Lx, t); Q.Ux = bv(PB.Ux, t); Q.Dinv = bv(PB.Dinv, t);
    Q.rhs = bv(PB.rhs, t); Q.sol = bv(PB.sol, t); Q.res = bv(PB.res, t); Q.cor = bv(PB.cor, t); Q.tmp = bv(PB.tmp, t);
    Q.part = bv(PB.part, t);
    Q.lam = bv(PB.lam, t); Q.wsc = bv(PB.wsc, t); Q.ds = bv(PB.ds, t); Q.dz = bv(PB.dz, t); Q.corr = bv(PB.corr, t);
    Q.rz = bv(PB.rz, t); Q.eta = bv(PB.eta, t); Q.rx = bv(PB.rx, t); Q.ry = bv(PB.ry, t);
    Solver<Ctx> sv(S, Q, O, cx);
#ifdef CONIC_PROF
    const long long tall_ = (long long)wall_clock64();
#endif
    const Result R = sv.run(live);
#ifdef CONIC_PROF
    if (live && cx.w == 0 && t == 0) {
        const double us = 1e-2;      // wall_clock64: 100 MHz
        printf("CONIC_PROF problem 0: iters %d | total %.0f us | factor %.0f fwd %.0f bwd %.0f residual %.0f (rows %.0f) scaling+Gt %.0f | solves %lld residuals %lld\n",
               R.iters, us * ((long long)wall_clock64() - tall_), us * sv.prof_[0], us * sv.prof_[1], us * sv.prof_[2], us * sv.prof_[3], us * sv.prof_[5], us * sv.prof_[4], sv.prof_[6], sv.prof_[7]);
    }
#endif
    if (!live || cx.w != 0) return;
    status[t] = R.status;
    iters[t] = R.iters;
    double* io = info + t;
    io[0 * info_es] = R.pcost; io[1 * info_es] = R.dcost; io[2 * info_es] = R.gap; io[3 * info_es] = R.pres;
    io[4 * info_es] = R.dres; io[5 * info_es] = R.relgap; io[6 * info_es] = (double)R.nreg; io[7 * info_es] = (double)R.nrefine;
}

// 64 x 64 tile transpose through LDS: both the [len, B] (len fastest) and the interleaved [len][BS] (problem fastest)
// side are accessed with coalesced rows.
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ src, double* __restrict__ dst, long rows,
                                                        long cols, long src_ld, long dst_ld)
{
    // src: rows x cols with leading dimension src_ld (element (r, c) at src[c * src_ld + r]);
    // dst: element (r, c) at dst[r * dst_ld + c]
    __shared__ double tile[64][65];
    const long r0 = (long)blockIdx.x * 64, c0 = (long)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) {
        const long r = r0 + tx, c = c0 + j;
        if (r < rows && c < cols) tile[j][tx] = src[c * src_ld + r];
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        const long r = r0 + j, c = c0 + tx;
        if (r < rows && c < cols) dst[r * dst_ld + c] = tile[tx][j];
    }
}

int transpose_to_interleaved(hipStream_t st, const double* src, double* dst, long len, int B, int BS)
{
    if (len <= 0 || B <= 0) return SCP_OK;
    dim3 grid((unsigned)((len + 63) / 64), (unsigned)((B + 63) / 64));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, st, src, dst, len, (long)B, len, (long)BS);
    return hipGetLastError() == hipSuccess ? SCP_OK : SCP_ERR_HIP;
}
int transpose_from_interleaved(hipStream_t st, const double* src, double* dst, long len, int B, int BS)
{
    if (len <= 0 || B <= 0) return SCP_OK;
    // src element (t, e) at src[e * BS + t] : rows = B (fastest), cols = len ; dst element (t, e) at dst[t * len + e]
    dim3 grid((unsigned)((B + 63) / 64), (unsigned)((len + 63) / 64));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, st, src, dst, (long)B, len, (long)BS, len);
    return hipGetLastError() == hipSuccess ? SCP_OK : SCP_ERR_HIP;
}

#define ENG_TRY(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return SCP_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

template <class T>
static int upload(Engine& E, const std::vector<T>& v, const T** out)
{
    void* d = nullptr;
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    if (hipMalloc(&d, bytes) != hipSuccess) { E.err = "hipMalloc (schedule)"; return SCP_ERR_ALLOC; }
    E.allocs.push_back(d);
    if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { E.err = "hipMemcpy (schedule)"; return SCP_ERR_HIP; }
    *out = (const T*)d;
    return SCP_OK;
}

// the factorisation part of a schedule (everything that depends on the ordering)
static int upload_factor_schedule(Engine& E, const Symbolic& S, Sched& D)
{
    std::vector<int2_> pairs(S.pair_a.size());
    for (size_t i = 0; i < pairs.size(); i++) { pairs[i].a = S.pair_a[i]; pairs[i].b = S.pair_b[i]; }
    std::vector<long long> pair_p(S.pair_p.begin(), S.pair_p.end());
    int rc;
#define UP(vec, field) if ((rc = upload(E, vec, &D.field)) != SCP_OK) return rc
    D.nnzL = S.Lp[S.nk];
    UP(S.perm, perm);
    UP(S.Lp, Lp); UP(S.Li, Li); UP(S.l_src, l_src); UP(S.l_src_idx, l_src_idx); UP(S.d_src, d_src); UP(S.d_src_idx, d_src_idx);
    UP(S.d_kind, d_kind);
    UP(pair_p, pair_p); UP(pairs, pairs);
    UP(S.row_p, row_p); UP(S.row_k, row_k); UP(S.row_pos, row_pos);
    D.nlev = (int)S.lev_p.size() - 1; D.nrlev = (int)S.rlev_p.size() - 1;
    UP(S.lev_p, lev_p); UP(S.lev_cols, lev_cols); UP(S.lev_ent_p, lev_ent_p); UP(S.lev_ent, lev_ent); UP(S.ent_col, ent_col);
    UP(S.rlev_p, rlev_p); UP(S.rlev_cols, rlev_cols);
    std::vector<long long> q0(S.echunk_q0.begin(), S.echunk_q0.end()), q1(S.echunk_q1.begin(), S.echunk_q1.end());
    D.max_chunks = S.max_chunks;
    UP(S.lev_nshort, lev_nshort); UP(S.rchunk_p, rchunk_p); UP(S.rchunk_r0, rchunk_r0); UP(S.rchunk_r1, rchunk_r1);
    UP(S.col_c0, col_c0); UP(S.col_c1, col_c1);
    UP(S.lev_ent_nshort, lev_ent_nshort); UP(S.echunk_p, echunk_p); UP(S.ent_c0, ent_c0); UP(S.ent_c1, ent_c1);
    UP(q0, echunk_q0); UP(q1, echunk_q1);
#undef UP
    return SCP_OK;
}

// problems per wave = 64 / SUB by batch size
// (round 5: one workgroup per problem up to 768 problems, 320 before -- with the cooperative item groups that geometry solves the
//  free-flyer N = 200 program at batch 512 in 3.9 s per launch against 7.0 s with 16 sub-workers; equal at 1 024, slower at 2 048)
static int default_sub_workers(int B) { return B >= 12288 ? 1 : (B >= 2048 ? 4 : (B > 768 ? 16 : 64)); }

int Engine::create(int n, int p, int m, int l, const std::vector<int>& q_in, const Csc& P, const Csc& A, const Csc& G,
                   const int* perm, int capacity, int dev)
{
    if (capacity < 1) { err = "batch_capacity < 1"; return SCP_ERR_BAD_ARGUMENT; }
    // q[c] > 0: second-order cone of that dimension; q[c] = -3: exponential cone (include/scp_conic.h).  For the symbolic
    // analysis an exponential cone is a dense 3-row block like a second-order cone of dimension 3.
    std::vector<int> q(q_in), ctype(q_in.size(), 0), cexp(q_in.size(), -1);
    int nexp = 0;
    for (size_t c = 0; c < q.size(); c++) {
        if (q[c] == -3) { q[c] = 3; ctype[c] = 1; cexp[c] = nexp++; }
        else if (q[c] < 1) { err = "cone dimension < 1 (an exponential cone is q = -3)"; return SCP_ERR_BAD_ARGUMENT; }
    }
    const char* om = std::getenv("SCP_CONIC_ORDER");
    const std::string order = om ? om : "auto";
    const bool free_order = std::getenv("SCP_CONIC_FREE_ORDER") != nullptr;
    try {
        // Round 2 kept pure LPs (Starship: no second-order cone, no quadratic cost) on the sequential order: their node
        // blocks P + Gt'Gt are numerically singular late in a run and the nested order then broke down in overflowing pivots
        // after a dynamic regularisation.  Since the solver repeats such a factorisation with a larger static regularisation
        // (conic_ipm.hpp, run()) the nested order carries those programs too: Starship SCvx N = 100, 1 164 -> 118 levels, the
        // same iteration counts and optima as the sequential order on successive subproblems (tests/test_template_cpu.py).
        // Which order: the cheaper schedule of the two by schedule_cost (conic_symbolic.hpp, analyse_auto) -- the nested order
        // only pays where the dissection finds the chain (quadrotor GuSTO N = 30, slack-free form: it did not with the first
        // threshold and the run cost 3 x the sequential schedule; measured in profiles/README.md, round 3).
        const bool try_nd = perm == nullptr && !free_order && order != "seq";
        if (try_nd) {
            bool nested = false;
            const int waves = waves_per_group < 1 ? 1 : (waves_per_group > CONIC_MAX_WAVES ? CONIC_MAX_WAVES : waves_per_group);
            const int workers = waves * (sub_workers > 0 ? sub_workers : default_sub_workers(capacity));   // lanes per problem
            sym = analyse_auto(n, p, m, l, q, P, A, G, workers, order != "nd", &sym_fb, &nested);
            has_fb = nested && order != "nd";
        } else {
            sym = analyse(n, p, m, l, q, P, A, G, perm, free_order, ORDER_SEQUENTIAL);
            has_fb = false;
        }
    } catch (const std::exception& e) {
        err = e.what();
        return SCP_ERR_BAD_ARGUMENT;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { err = "no HIP device"; return SCP_ERR_NO_DEVICE; }
    if (dev < 0 || dev >= ndev) { err = "device ordinal out of range"; return SCP_ERR_BAD_ARGUMENT; }
    device = dev;
    ENG_TRY(hipSetDevice(dev));
    cap = capacity; BS = (capacity + 63) & ~63;
    const Symbolic& S = sym;
    Sched& D = sched;
    D.n = n; D.p = p; D.m = m; D.l = l; D.nk = S.nk; D.ncone = (int)q.size(); D.nexp = nexp;
    D.nnzG = G.nnz(); D.nnzGt = S.Gt.nnz(); D.nnzA = A.nnz(); D.nnzP = P.nnz();
    D.njob = (int)S.job_gt0.size(); D.nlp = (int)S.lp_gt.size();
    const CsrView Gr = csr_view(G);
    int rc;
#define UP(vec, field) if ((rc = upload(*this, vec, &D.field)) != SCP_OK) return rc
    UP(S.q, q); UP(S.cone_off, cone_off); UP(ctype, ctype); UP(cexp, cexp);
    UP(G.p, Gp); UP(G.i, Gi); UP(Gr.p, Gr_p); UP(Gr.j, Gr_j); UP(Gr.pos, Gr_pos);
    UP(S.Gt.p, Gtp); UP(S.Gt.i, Gti); UP(S.Gtr.p, Gtr_p); UP(S.Gtr.j, Gtr_j); UP(S.Gtr.pos, Gtr_pos);
    UP(A.p, Ap); UP(A.i, Ai); UP(S.Ar.p, Ar_p); UP(S.Ar.j, Ar_j); UP(S.Ar.pos, Ar_pos);
    UP(S.Pfull.p, Pf_p); UP(S.Pfull.j, Pf_j); UP(S.Pfull.pos, Pf_pos);
    UP(S.kk_p, kk_p); UP(S.kk_src, kk_src); UP(S.kk_idx, kk_idx); UP(S.kk_col, kk_col);
    UP(S.kk_long, kk_long); D.nkk_long = (int)S.kk_long.size(); D.kk_long_thr = Symbolic::KK_LONG;
    UP(S.job_gt0, job_gt0); UP(S.job_cone, job_cone); UP(S.job_src_p, job_src_p); UP(S.job_src_row, job_src_row);
    UP(S.job_src_g, job_src_g); UP(S.lp_gt, lp_gt); UP(S.lp_g, lp_g);
#undef UP
    if ((rc = upload_factor_schedule(*this, S, D)) != SCP_OK) return rc;
    long nnzL_max = D.nnzL, chunks_max = D.max_chunks;
    if (has_fb) {     // same program, same pattern arrays; only the factorisation schedule differs
        // the sequential schedule is built and uploaded only for the diagnostic mode that uses it (SCP_CONIC_FALLBACK=seq);
        // the default second attempt runs on the primary schedule with a larger static regularisation (launch())
        const char* fbm = std::getenv("SCP_CONIC_FALLBACK");
        if (fbm && std::string(fbm) == "seq") {
            sched_fb = D;
            if ((rc = upload_factor_schedule(*this, sym_fb, sched_fb)) != SCP_OK) return rc;
            nnzL_max = std::max<long>(nnzL_max, sched_fb.nnzL);
            chunks_max = std::max<long>(chunks_max, sched_fb.max_chunks);
        } else {
            sym_fb = Symbolic();      // (the host copy is not needed in the default mode)
        }
    }
    {   // mask / counter of the further attempts (launch()): for every schedule since round 5
        void* dm = nullptr;
        if (hipMalloc(&dm, sizeof(int) * ((size_t)BS + 1)) != hipSuccess) { err = "hipMalloc (fallback mask)"; return SCP_ERR_ALLOC; }
        allocs.push_back(dm);
        fb_mask = (int*)dm; fb_count = fb_mask + BS;
    }
    // ---- buffers ----
    auto dalloc = [&](double** ptr, long len) -> int {
        void* d = nullptr;
        const size_t bytes = (size_t)std::max<long>(len, 1) * sizeof(double);
        if (hipMalloc(&d, bytes) != hipSuccess) { err = "hipMalloc (buffers)"; return SCP_ERR_ALLOC; }
        allocs.push_back(d);
        *ptr = (double*)d;
        return SCP_OK;
    };
    const long nnzG = D.nnzG, nnzA = D.nnzA, nnzP = D.nnzP, nnzGt = D.nnzGt, nnzL = nnzL_max, nk = D.nk, nc = D.ncone;
#define DA(ptr, len) if ((rc = dalloc(&ptr, (len))) != SCP_OK) return rc
    DA(c, (long)n * BS); DA(b, (long)p * BS); DA(h, (long)m * BS); DA(Gx, nnzG * BS); DA(Ax, nnzA * BS); DA(Px, nnzP * BS);
    DA(c_sh, n); DA(b_sh, p); DA(h_sh, m); DA(Gx_sh, nnzG); DA(Ax_sh, nnzA); DA(Px_sh, nnzP);
    DA(x, (long)n * BS); DA(y, (long)p * BS); DA(z, (long)m * BS); DA(s, (long)m * BS);
    const long work_len = nnzGt + 2 * nnzL + nk + 5 * nk + chunks_max + 6 * (long)m + nc + 9L * nexp + n + p;
    DA(work, work_len * BS);
    DA(info, 8L * BS);
#undef DA
    void* d = nullptr;
    if (hipMalloc(&d, sizeof(int) * 2 * BS) != hipSuccess) { err = "hipMalloc (status)"; return SCP_ERR_ALLOC; }
    allocs.push_back(d);
    status = (int*)d; iters = status + BS;
    bytes_per_problem = 8LL * (2 * (n + p + 2L * m) + nnzG + nnzA + nnzP + work_len + 8);
    return SCP_OK;
}

void Engine::destroy()
{
    for (void* p_ : allocs) (void)hipFree(p_);
    allocs.clear();
}

// mask[t] = 1 for the problems of the primary pass that must be re-solved with the sequential schedule
__global__ void fallback_mask_kernel(const int* status, const int* active, int* mask, int* count, int B)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B) return;
    const bool live = active == nullptr || active[t] != 0;
    const int st = status[t];
    // only UNSAFE exits are re-solved: ALMOST_OPTIMAL is a usable solution for the SCP loops (unsafe_solution, scp.jl:965-980)
    // and the sequential pass overwrites x / y / z / s -- re-solving it could turn a usable solution into a failure
    const int need = live && (st == ST_ITERLIM || st == ST_NUMERR) ? 1 : 0;
    mask[t] = need;
    if (need) atomicAdd(count, 1);
}

// count[0] += problems of the fallback pass that left it with a usable solution or a certificate
__global__ void fallback_rescued_kernel(const int* status, const int* mask, int* count, int B)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B || !mask[t]) return;
    const int st = status[t];
    if (st != ST_ITERLIM && st != ST_NUMERR) atomicAdd(count, 1);
}

static int launch_one(Engine& E, const Sched& D, hipStream_t stream, int B, const Opts& oe, unsigned shared_mask, const int* active)
{
    const int BS = E.BS;
    const int waves = E.waves_per_group < 1 ? 1 : (E.waves_per_group > CONIC_MAX_WAVES ? CONIC_MAX_WAVES : E.waves_per_group);
    ProbBase PB;
    auto il = [&](double* ptr) { return Arr{ptr, (long)BS, 1}; };
    auto sh = [&](double* ptr) { return Arr{ptr, 1, 0}; };
    PB.c = (shared_mask & SCP_CONIC_SHARED_C) ? sh(E.c_sh) : il(E.c);
    PB.b = (shared_mask & SCP_CONIC_SHARED_B) ? sh(E.b_sh) : il(E.b);
    PB.h = (shared_mask & SCP_CONIC_SHARED_H) ? sh(E.h_sh) : il(E.h);
    PB.Gx = (shared_mask & SCP_CONIC_SHARED_G) ? sh(E.Gx_sh) : il(E.Gx);
    PB.Ax = (shared_mask & SCP_CONIC_SHARED_A) ? sh(E.Ax_sh) : il(E.Ax);
    PB.Px = (shared_mask & SCP_CONIC_SHARED_P) ? sh(E.Px_sh) : il(E.Px);
    PB.x = il(E.x); PB.y = il(E.y); PB.z = il(E.z); PB.s = il(E.s);
    // sub-workers per wave: small batches are spread over more workgroups (problems per wave 64 / SUB)
    int sub = E.sub_workers;
    // measured on the rocket program (profiles/README.md); batches of a few hundred problems give every problem a whole
    // workgroup (SUB = 64: 1024 workers per problem, one workgroup per CU)
    if (sub <= 0) sub = default_sub_workers(B);
    // Work arrays (scaled G, the factor L / U = L D, right-hand sides, ...): interleaved across the batch [element][BS] when a
    // wave holds several problems (a load of "element e" is one contiguous segment), but PROBLEM-MAJOR [problem][element] when
    // a workgroup owns one problem (SUB = 64): its 1024 workers then gather inside that problem's own contiguous factor instead
    // of touching one 8-byte word per 128-byte line of the interleaved array (16x read amplification, measured: Starship
    // N = 100 at 256 problems was bandwidth-bound on bytes it never used).  The arrays live and die inside one launch.
    const bool problem_major = sub == 64;
    double* w = E.work;
    auto take = [&](long len) { Arr a = problem_major ? Arr{w, 1, len} : il(w); w += len * BS; return a; };
    PB.Gt = take(D.nnzGt); PB.Lx = take(D.nnzL); PB.Ux = take(D.nnzL); PB.Dinv = take(D.nk);
    PB.rhs = take(D.nk); PB.sol = take(D.nk); PB.res = take(D.nk); PB.cor = take(D.nk); PB.tmp = take(D.nk);
    PB.part = take(D.max_chunks);
    PB.lam = take(D.m); PB.wsc = take(D.m); PB.ds = take(D.m); PB.dz = take(D.m); PB.corr = take(D.m); PB.rz = take(D.m);
    PB.eta = take(D.ncone + 9L * D.nexp); PB.rx = take(D.n); PB.ry = take(D.p);
    const int ppw = 64 / sub;
    const dim3 grid((B + ppw - 1) / ppw), block(64 * waves);
    int* status = E.status; int* iters = E.iters; double* info = E.info;
#define CONIC_LAUNCH(MAXW, SUB) \
    hipLaunchKernelGGL((conic_ipm_kernel<MAXW, SUB>), grid, block, 0, stream, D, PB, oe, B, active, status, iters, info, (long)BS)
    if (waves > 8) {
        if (sub == 1) CONIC_LAUNCH(16, 1); else if (sub == 4) CONIC_LAUNCH(16, 4); else if (sub == 16) CONIC_LAUNCH(16, 16); else CONIC_LAUNCH(16, 64);
    } else {
        if (sub == 1) CONIC_LAUNCH(8, 1); else if (sub == 4) CONIC_LAUNCH(8, 4); else if (sub == 16) CONIC_LAUNCH(8, 16); else CONIC_LAUNCH(8, 64);
    }
#undef CONIC_LAUNCH
    return hipGetLastError() == hipSuccess ? SCP_OK : SCP_ERR_HIP;
}

int Engine::launch(hipStream_t stream, int B, const Opts& o, unsigned shared_mask, const int* active)
{
    if (B < 1 || B > cap) { err = "batch size exceeds batch_capacity"; return SCP_ERR_BATCH_TOO_LARGE; }
    Opts oe = o;
    oe.fine = 0;
    if (!(oe.reg >= 0.0)) { oe.reg = auto_reg(sym.n_free, sym.n, sym.m, sym.q.empty() && sym.P.i.empty(), sched.nexp > 0); oe.fine = oe.reg < 1e-9; }
    n_launched += B;
    if (launch_one(*this, sched, stream, B, oe, shared_mask, active) != SCP_OK) { err = "conic_ipm_kernel launch failed"; return SCP_ERR_HIP; }
    if (!fb_mask) return SCP_OK;
    // (the count of unusable exits is read back after every launch: one 4-byte copy and a stream synchronisation per conic launch --
    // the callers synchronise on the same stream right after, for the solution they asked for)
    // ---- further attempts for the problems that ended ITERATION_LIMIT / NUMERICAL_ERROR (ALMOST_OPTIMAL is usable and kept) ----
    // On the SAME schedule with a larger static regularisation: what fails on the degenerate LPs of the Starship and on GuSTO
    // subproblems whose penalty weight has escalated is a rounding lottery of the factorisation (wrong-signed pivots from cancelling
    // sums -> dynamic regularisations), not a property of the elimination order -- the same instance in another summation order is
    // solved.  Round 5: TWO further attempts, 10 x and 100 x the first one's regularisation (round 4: one, at
    // 100 x): quadrotor GuSTO instance 21 at lambda = 1.25e6 ends ITERATION_LIMIT / ALMOST_OPTIMAL with 51 dynamic regularisations
    // at 1e-8 depending on the order of the sums, OPTIMAL in 28 iterations with 2 at 1e-7 in EVERY order, and stalls at 1e-6; and
    // for every schedule, not only the nested ones (the attempt no longer needs a second schedule).  A problem's attempts depend on
    // ITS OWN exits only (batch independence).  SCP_CONIC_FALLBACK=seq: the sequential schedule as the one further attempt.
    static const bool fb_seq = std::getenv("SCP_CONIC_FALLBACK") && std::string(std::getenv("SCP_CONIC_FALLBACK")) == "seq";
    // The ladder climbs from the attempt before it: r1 = max(10 reg, 1e-7), r2 = min(10 r1, 1e-4) -- with 10 x / 100 x of the
    // FIRST value both clamped to the 1e-7 floor, the 1e-10 class repeated the same solve twice and never reached 1e-6.
    double reg_prev = oe.reg;
    for (int att = 0; att < (fb_seq ? 1 : 2); att++) {
        ENG_TRY(hipMemsetAsync(fb_count, 0, sizeof(int), stream));
        hipLaunchKernelGGL(fallback_mask_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, status, active, fb_mask, fb_count, B);
        int nfb = 0;
        ENG_TRY(hipMemcpyAsync(&nfb, fb_count, sizeof(int), hipMemcpyDeviceToHost, stream));
        ENG_TRY(hipStreamSynchronize(stream));
        if (nfb == 0) return SCP_OK;
        if (att == 0) n_fallback += nfb;
        Opts o2 = oe;
        if (!fb_seq) {
            o2.reg = std::min(std::max(reg_prev * 10.0, 1e-7), 1e-4);
            if (o2.reg <= reg_prev) return SCP_OK;      // nothing left to try (the previous attempt already ran at the cap)
            reg_prev = o2.reg;
        }
        if (fb_seq && !has_fb) return SCP_OK;
        if (launch_one(*this, fb_seq ? sched_fb : sched, stream, B, o2, shared_mask, fb_mask) != SCP_OK) { err = "conic_ipm_kernel (further attempt) launch failed"; return SCP_ERR_HIP; }
        // what did the attempt buy?  (a problem is rescued when it now holds a usable solution or a certificate)
        ENG_TRY(hipMemsetAsync(fb_count, 0, sizeof(int), stream));
        hipLaunchKernelGGL(fallback_rescued_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, status, fb_mask, fb_count, B);
        int nres = 0;
        ENG_TRY(hipMemcpyAsync(&nres, fb_count, sizeof(int), hipMemcpyDeviceToHost, stream));
        ENG_TRY(hipStreamSynchronize(stream));
        n_rescued += nres;
        // Only the diagnostic sequential mode still adopts the sequential schedule when that is what rescues a launch.
        if (fb_seq && 4L * nfb > B && 2L * nres > nfb) { sched = sched_fb; sym = sym_fb; has_fb = false; }
        if (nres == nfb) return SCP_OK;
    }
    return SCP_OK;
}

}  // namespace conic
}  // namespace scp

// ------------------------------------------------------------------------------------------------------------------
struct scp_conic {
    Engine eng;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double* stage = nullptr;   // staging buffer for the [len, B] host layout
    long stage_len = 0;
    std::string err;
};

extern "C" void scp_conic_default_opts(scp_conic_opts* o)
{
    if (!o) return;
    const Opts d = default_opts();
    o->max_iter = d.max_iter; o->feastol = d.feastol; o->abstol = d.abstol; o->reltol = d.reltol; o->reg = d.reg;
    o->dyn_eps = d.dyn_eps; o->dyn_delta = d.dyn_delta; o->nref = d.nref; o->ref_tol = d.ref_tol; o->step = d.step;
}

extern "C" const char* scp_conic_last_error(scp_conic_handle h) { return h ? h->err.c_str() : "null handle"; }

static Csc make_csc(int nrow, int ncol, const int* p, const int* i)
{
    Csc M;
    M.nrow = nrow; M.ncol = ncol;
    if (p == nullptr) { M.p.assign(ncol + 1, 0); return M; }
    M.p.assign(p, p + ncol + 1);
    const int nnz = M.p[ncol] > 0 ? M.p[ncol] : 0;
    if (nnz > 0 && i != nullptr) M.i.assign(i, i + nnz);
    return M;
}

extern "C" int scp_conic_create(int n, int p, int m, int l, int ncones, const int* q, const int* Pp, const int* Pi,
                                const int* Ap, const int* Ai, const int* Gp, const int* Gi, const int* perm,
                                int batch_capacity, int device, scp_conic_handle* out)
{
    if (!out) return SCP_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (n < 1 || p < 0 || m < 0 || l < 0 || ncones < 0 || (ncones > 0 && !q)) return SCP_ERR_BAD_ARGUMENT;
    scp_conic* h = new (std::nothrow) scp_conic;
    if (!h) return SCP_ERR_ALLOC;
    std::vector<int> qv(q, q + ncones);
    int rc = h->eng.create(n, p, m, l, qv, make_csc(n, n, Pp, Pi), make_csc(p, n, Ap, Ai), make_csc(m, n, Gp, Gi), perm,
                           batch_capacity, device);
    if (rc == SCP_OK) {
        if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
            hipEventCreate(&h->ev1) != hipSuccess) rc = SCP_ERR_HIP;
    }
    if (rc == SCP_OK) {
        const Sched& D = h->eng.sched;
        long mx = std::max<long>(std::max<long>(D.nnzG, D.nnzA), std::max<long>(D.nnzP, std::max<long>(D.m, std::max<long>(D.n, D.p))));
        mx = std::max<long>(mx, 8);   // info[8, B] goes through the same staging buffer
        h->stage_len = mx * h->eng.cap;
        if (hipMalloc((void**)&h->stage, sizeof(double) * std::max<long>(h->stage_len, 1)) != hipSuccess) rc = SCP_ERR_ALLOC;
    }
    if (rc != SCP_OK) {
        // creation failed: report through the return code only (the handle is not handed out)
        h->eng.destroy();
        if (h->stage) (void)hipFree(h->stage);
        if (h->ev0) (void)hipEventDestroy(h->ev0);
        if (h->ev1) (void)hipEventDestroy(h->ev1);
        if (h->stream) (void)hipStreamDestroy(h->stream);
        delete h;
        return rc;
    }
    *out = h;
    return SCP_OK;
}

extern "C" int scp_conic_destroy(scp_conic_handle h)
{
    if (!h) return SCP_ERR_BAD_ARGUMENT;
    (void)hipSetDevice(h->eng.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->eng.destroy();
    if (h->stage) (void)hipFree(h->stage);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return SCP_OK;
}

extern "C" int scp_conic_stats(scp_conic_handle h, long long stats[16])
{
    if (!h || !stats) return SCP_ERR_BAD_ARGUMENT;
    stats[0] = h->eng.sched.nnzL; stats[1] = h->eng.sym.flops; stats[2] = h->eng.sched.nk; stats[3] = h->eng.sched.nnzGt;
    stats[4] = h->eng.bytes_per_problem;
    stats[5] = h->eng.sched.nlev; stats[6] = h->eng.sched.nrlev; stats[7] = h->eng.waves_per_group;
    stats[8] = h->eng.sym.nd_depth; stats[9] = h->eng.n_fallback; stats[10] = h->eng.n_launched;
    stats[11] = h->eng.has_fb ? h->eng.sched_fb.nlev : 0;
    stats[12] = h->eng.n_rescued; stats[13] = stats[14] = stats[15] = 0;
    return SCP_OK;
}

#define CH_TRY(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            h->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return SCP_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

static int put_array(scp_conic* h, const double* src, double* dst_il, double* dst_sh, long len, int B, bool shared)
{
    if (len == 0) return SCP_OK;
    if (!src) { h->err = "missing input array"; return SCP_ERR_BAD_ARGUMENT; }
    if (shared) {
        CH_TRY(hipMemcpyAsync(dst_sh, src, sizeof(double) * len, hipMemcpyHostToDevice, h->stream));
        return SCP_OK;
    }
    CH_TRY(hipMemcpyAsync(h->stage, src, sizeof(double) * len * B, hipMemcpyHostToDevice, h->stream));
    const int rc = transpose_to_interleaved(h->stream, h->stage, dst_il, len, B, h->eng.BS);
    if (rc != SCP_OK) h->err = "transpose launch failed";
    return rc;
}
static int get_array(scp_conic* h, const double* src_il, double* dst, long len, int B)
{
    if (len == 0 || !dst) return SCP_OK;
    const int rc = transpose_from_interleaved(h->stream, src_il, h->stage, len, B, h->eng.BS);
    if (rc != SCP_OK) { h->err = "transpose launch failed"; return rc; }
    CH_TRY(hipMemcpyAsync(dst, h->stage, sizeof(double) * len * B, hipMemcpyDeviceToHost, h->stream));
    CH_TRY(hipStreamSynchronize(h->stream));   // the staging buffer is reused by the next array
    return SCP_OK;
}

extern "C" int scp_conic_solve_batch_host(scp_conic_handle h, int B, const double* c, const double* b, const double* hvec,
                                          const double* Gx, const double* Ax, const double* Px, unsigned shared_mask,
                                          const scp_conic_opts* opts, double* x, double* y, double* z, double* s,
                                          int32_t* status, int32_t* iters, double* info, double* seconds)
{
    if (!h) return SCP_ERR_BAD_ARGUMENT;
    Engine& E = h->eng;
    if (B < 1) { h->err = "B < 1"; return SCP_ERR_BAD_ARGUMENT; }
    if (B > E.cap) { h->err = "batch size exceeds batch_capacity"; return SCP_ERR_BATCH_TOO_LARGE; }
    CH_TRY(hipSetDevice(E.device));
    const Sched& D = E.sched;
    Opts o = default_opts();
    if (opts) {
        if (opts->max_iter < 0 || opts->nref < 0) { h->err = "bad solver options"; return SCP_ERR_BAD_ARGUMENT; }
        o.max_iter = opts->max_iter; o.feastol = opts->feastol; o.abstol = opts->abstol; o.reltol = opts->reltol;
        o.reg = opts->reg; o.dyn_eps = opts->dyn_eps; o.dyn_delta = opts->dyn_delta; o.nref = opts->nref;
        o.ref_tol = opts->ref_tol; o.step = opts->step;
    }
    int rc;
    // every put reuses the staging buffer: stream order keeps them apart
    if ((rc = put_array(h, c, E.c, E.c_sh, D.n, B, shared_mask & SCP_CONIC_SHARED_C)) != SCP_OK) return rc;
    if ((rc = put_array(h, b, E.b, E.b_sh, D.p, B, shared_mask & SCP_CONIC_SHARED_B)) != SCP_OK) return rc;
    if ((rc = put_array(h, hvec, E.h, E.h_sh, D.m, B, shared_mask & SCP_CONIC_SHARED_H)) != SCP_OK) return rc;
    if ((rc = put_array(h, Gx, E.Gx, E.Gx_sh, D.nnzG, B, shared_mask & SCP_CONIC_SHARED_G)) != SCP_OK) return rc;
    if ((rc = put_array(h, Ax, E.Ax, E.Ax_sh, D.nnzA, B, shared_mask & SCP_CONIC_SHARED_A)) != SCP_OK) return rc;
    if ((rc = put_array(h, Px, E.Px, E.Px_sh, D.nnzP, B, shared_mask & SCP_CONIC_SHARED_P)) != SCP_OK) return rc;
    CH_TRY(hipEventRecord(h->ev0, h->stream));
    if ((rc = E.launch(h->stream, B, o, shared_mask)) != SCP_OK) { h->err = E.err; return rc; }
    CH_TRY(hipEventRecord(h->ev1, h->stream));
    CH_TRY(hipStreamSynchronize(h->stream));
    if (seconds) { float ms = 0; CH_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1)); *seconds = ms * 1e-3; }
    if ((rc = get_array(h, E.x, x, D.n, B)) != SCP_OK) return rc;
    if ((rc = get_array(h, E.y, y, D.p, B)) != SCP_OK) return rc;
    if ((rc = get_array(h, E.z, z, D.m, B)) != SCP_OK) return rc;
    if ((rc = get_array(h, E.s, s, D.m, B)) != SCP_OK) return rc;
    if ((rc = get_array(h, E.info, info, 8, B)) != SCP_OK) return rc;
    if (status) CH_TRY(hipMemcpy(status, E.status, sizeof(int) * B, hipMemcpyDeviceToHost));
    if (iters) CH_TRY(hipMemcpy(iters, E.iters, sizeof(int) * B, hipMemcpyDeviceToHost));
    return SCP_OK;
}

extern "C" int socp_solve_batch(int n, int m, int p, int l, int ncones, const int* q, const int* Gp, const int* Gi,
                                const double* Gx, const int* Ap, const int* Ai, const double* Ax, const double* c,
                                const double* hvec, const double* b, int B, double* x, double* y, double* s, double* z,
                                int32_t* status)
{
    scp_conic_handle h = nullptr;
    int rc = scp_conic_create(n, p, m, l, ncones, q, nullptr, nullptr, Ap, Ai, Gp, Gi, nullptr, B, 0, &h);
    if (rc != SCP_OK) return rc;
    rc = scp_conic_solve_batch_host(h, B, c, b, hvec, Gx, Ax, nullptr, 0u, nullptr, x, y, z, s, status, nullptr, nullptr, nullptr);
    scp_conic_destroy(h);
    return rc;
}
