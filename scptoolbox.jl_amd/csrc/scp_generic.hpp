// Generic SCP subproblem pipeline on the device (included at the end of scp_api.hip): `solve_subproblem!` for ANY
// subproblem the host formulated as a conic template (scptoolbox.jl_amd/subproblem.py: PTR with q_tr in {1, 2, 4, Inf},
// SCvx, correct_convex!, ...) -- the reference's per-iteration JuMP formulation (src/solvers/ptr.jl:213-293,
// scvx.jl:225-303, scp.jl:657-895) becomes ONE gather kernel:
//
//   reference trajectories (resident) --discretize! (K1)--> ref.dyn
//        --linearise (s, C, D, G and the boundary conditions about the reference; per model)-->  source vector
//        --affine gather  value[slot] = const[slot] + sum coef * src[.]  -->  c, b, h, Gx, Ax, Px  (interleaved)
//        --conic_ipm_kernel (conic_api.hip)--> conic solution --read-out--> x, u, p (un-scaled), linear functionals
//        --discretize! (K1) of the new point--> defects, feasibility            (scp.jl:169-181 couples them)
//
// plus the SCvx outer loop on the device (update rule, accept / reject, trust-region radius: scvx.jl:711-770,
// 924-1045).  The specialised stage-structured path (stage_problem.hpp + ipm2_*.hpp) stays the fast path for PTR with
// q_tr = Inf.
//
// Source vector layout (interleaved [nsrc][BS], every segment in Julia / column-major order), shared with
// subproblem.py::standard_sources:
//   xref(nx,N) uref(nu,N) pref(np + np_node N) A(nx,nx,N-1) Bm(nx,nu,N-1) Bp(nx,nu,N-1) F(nx,npF,N-1) r(nx,N-1) E(nx,nx,N-1)
//   C(ns,nx,N) D(ns,nu,N) Gs(ns,np + np_node,N) rs(ns,N) H0(nic,nx) K0(nic,np) l0(nic) Hf(ntc,nx) Kf(ntc,np) lf(ntc) scal(nscal)
// (np = global parameters, np_node = parameters of one node: Gs holds the compact parameter Jacobian of s at its node)
#pragma once

#include "../../include/scp_conic.h"
#include "conic_engine.hpp"

namespace scp {

enum { SEG_XREF = 0, SEG_UREF, SEG_PREF, SEG_A, SEG_BM, SEG_BP, SEG_F, SEG_R, SEG_E, SEG_C, SEG_D, SEG_GS, SEG_RS, SEG_H0,
       SEG_K0, SEG_L0, SEG_HF, SEG_KF, SEG_LF, SEG_SCAL, SEG_COUNT };

struct SrcLayout {
    long off[SEG_COUNT + 1];
};
static SrcLayout src_layout(const scp_model_info& i, int N, int nscal)
{
    // parameter Jacobians: s sees the global parameters and its node's own (compact, npc columns); the boundary
    // conditions see the global parameters only (model_common.hpp)
    const long nx = i.nx, nu = i.nu, np = i.np, npt = i.np + (long)i.np_node * N, npc = i.np + i.np_node, npF = i.npF, ns = i.ns,
               nic = i.nic, ntc = i.ntc, M = N - 1;
    const long len[SEG_COUNT] = {nx * N, nu * N, npt, nx * nx * M, nx * nu * M, nx * nu * M, nx * npF * M, nx * M, nx * nx * M,
                                 ns * nx * N, ns * nu * N, ns * npc * N, ns * N, nic * nx, nic * np, nic, ntc * nx, ntc * np,
                                 ntc, nscal};
    SrcLayout L;
    L.off[0] = 0;
    for (int s = 0; s < SEG_COUNT; s++) L.off[s + 1] = L.off[s] + len[s];
    return L;
}

// ---- linearisation of the non-convex constraints and boundary conditions about the reference (scp.jl:744-895) ----
struct GenLinArgs {
    int B, N;
    long BS;
    const double *xd, *ud, *p, *pp;   // [nx,N,B], [nu,N,B], [np,B], [npp,B]
    double* src;                      // interleaved
    long oC, oD, oG, oRS, oH0, oK0, oL0, oHF, oKF, oLF;
    const int* active;
};
template <class M>
__global__ __launch_bounds__(256) void gen_linearise_kernel(GenLinArgs a, typename M::Params par)
{
    constexpr int nx = M::nx, nu = M::nu, np = M::np, npa = np > 0 ? np : 1, ns = M::ns, nsa = ns > 0 ? ns : 1, nic = M::nic,
                  ntc = M::ntc, nbc = nic > ntc ? nic : ntc, npc = np_compact<M>(), npca = npc > 0 ? npc : 1;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)a.B * (a.N + 1)) return;
    const int b = (int)(gid % a.B), k = (int)(gid / a.B);   // problem fastest: interleaved stores coalesce
    if (a.active != nullptr && a.active[b] == 0) return;
    const int N = a.N;
    const double* pr = a.p + (long)b * np_total<M>(N);
    auto put = [&](long e, double v) { a.src[e * a.BS + b] = v; };
    if (k < N) {
        if (ns == 0) return;
        const double* xk = a.xd + ((long)b * N + k) * nx;
        const double* uk = a.ud + ((long)b * N + k) * nu;
        const double tk = (1.0 - (double)k / (double)(N - 1)) * 0.0 + ((double)k / (double)(N - 1)) * 1.0;
        double s[nsa], C[nsa * nx], Dm[nsa * nu], G[nsa * npca];
        for (int i = 0; i < nsa * npca; i++) G[i] = 0.0;
        M::s_eval(par, tk, k + 1, xk, uk, pr, s, C, Dm, G);
        const double* pk = pr + np + (long)M::np_node * k;     // this node's own parameters
        for (int i = 0; i < ns; i++) {
            double rr = s[i];
            for (int j = 0; j < nx; j++) { rr -= C[i * nx + j] * xk[j]; put(a.oC + i + ns * (j + (long)nx * k), C[i * nx + j]); }
            for (int j = 0; j < nu; j++) { rr -= Dm[i * nu + j] * uk[j]; put(a.oD + i + ns * (j + (long)nu * k), Dm[i * nu + j]); }
            for (int j = 0; j < npc; j++) {
                rr -= G[i * npca + j] * (j < np ? pr[j] : pk[j - np]);
                put(a.oG + i + ns * (j + (long)npc * k), G[i * npca + j]);
            }
            put(a.oRS + i + (long)ns * k, rr);     // s - C x - D u - G p  (scp.jl:778-783)
        }
        return;
    }
    const double* pp = a.pp + (long)b * M::npp;
    for (int which = 0; which < 2; which++) {
        const int nb = which == 0 ? nic : ntc;
        const double* xb = which == 0 ? a.xd + (long)b * N * nx : a.xd + ((long)b * N + (N - 1)) * nx;
        double g[nbc], H[nbc * nx], K[nbc * npa];
        for (int i = 0; i < nbc * npa; i++) K[i] = 0.0;
        if (which == 0) M::bc_ic(par, xb, pr, pp, g, H, K);
        else M::bc_tc(par, xb, pr, pp, g, H, K);
        const long oH = which == 0 ? a.oH0 : a.oHF, oK = which == 0 ? a.oK0 : a.oKF, oL = which == 0 ? a.oL0 : a.oLF;
        for (int i = 0; i < nb; i++) {
            double l = g[i];
            for (int j = 0; j < nx; j++) { l -= H[i * nx + j] * xb[j]; put(oH + i + (long)nb * j, H[i * nx + j]); }
            for (int j = 0; j < np; j++) { l -= K[i * npa + j] * pr[j]; put(oK + i + (long)nb * j, K[i * npa + j]); }
            put(oL + i, l);                        // g - H x - K p  (scp.jl:826-829, 862-865)
        }
    }
}

// ---- value[slot] = const[slot] + sum coef * src[.]  (thread per (problem, slot); problems along the lanes) ----
struct GatherArgs {
    int B, len;
    long BS;
    const double* val0;
    const int* ptr;
    const int* sidx;
    const double* coef;
    const double* src;   // interleaved [.][BS]
    double* dst;         // interleaved [len][BS]
};
__global__ __launch_bounds__(256) void gen_gather_kernel(GatherArgs a)
{
    const int b = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slot = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (b >= a.B || slot >= a.len) return;
    double v = a.val0[slot];
    for (int t = a.ptr[slot]; t < a.ptr[slot + 1]; t++) v += a.coef[t] * a.src[(long)a.sidx[t] * a.BS + b];
    a.dst[(long)slot * a.BS + b] = v;
}

// ---- un-scaling read-out: out[e, b] = S[e % dim] * x[idx[e]] + c[e % dim]  (value(blk), block.jl:368-394) ----
struct ReadoutArgs {
    int B, len, dim;
    long BS;
    const int* idx;
    const double *S, *c;
    const double* x;     // interleaved conic solution
    double* out;         // [len, B] problem-major
    const int* active;
};
__global__ __launch_bounds__(256) void gen_readout_kernel(ReadoutArgs a)
{
    const int b = blockIdx.x * 64 + (threadIdx.x & 63);
    const int e = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (b >= a.B || e >= a.len) return;
    if (a.active != nullptr && a.active[b] == 0) return;
    a.out[(long)b * a.len + e] = a.S[e % a.dim] * a.x[(long)a.idx[e] * a.BS + b] + a.c[e % a.dim];
}

// ---- per-problem quantities of the new point the outer loops need (scvx.jl:924-984, scp.jl:617-643, 909-931) ----
//   post[0] = L   original cost phi(x_N, p) + trapz Gamma          (compute_original_cost)
//   post[1] = trapz_k (||defect_k||_1 + ||max(s_k, 0)||_1) + ||g_ic||_1 + ||g_tc||_1   (actual_cost_penalty! / lambda)
//   post[2] = deviation ||dp||_q + max_k ||dx_k||_q in scaled variables, q = q_exit      (solution_deviation, scp.jl:909-931)
//   post[3] = the same in the trust-region norm q = q_tr (q = 4: squared norms)           (GuSTO trust_region_cost(:nonconvex), :1172-1185)
struct PostArgs {
    double q_exit, q_tr;   // >= 1 or Inf
    int B, N;
    const double *xd, *ud, *p, *pp, *defect;   // the new point and its defects
    const double *rxd, *rp;                    // reference (deviation)
    const double *Sx, *Sp;
    double* post;                              // [B][4]
    const int* active;
};
// norm(v, q) of src/solvers/scp.jl:926-929 (LinearAlgebra.norm): q = Inf, 1, 2 or any q >= 1
template <int n>
__device__ __forceinline__ double qnorm_small(const double (&v)[n], double q)
{
    double acc = 0.0;
    if (isinf(q)) { for (int i = 0; i < n; i++) acc = fmax(acc, v[i]); return acc; }
    if (q == 1.0) { for (int i = 0; i < n; i++) acc += v[i]; return acc; }
    if (q == 2.0) { for (int i = 0; i < n; i++) acc += v[i] * v[i]; return sqrt(acc); }
    for (int i = 0; i < n; i++) acc += pow(v[i], q);
    return pow(acc, 1.0 / q);
}
// || (a - b) ./ S ||_q over len entries, computed by one wave (every lane returns the norm)
__device__ __forceinline__ double qnorm_wave(const double* a, const double* b, const double* S, int len, double q, int lane)
{
    double acc = 0.0;
    const bool inf = isinf(q);
    for (int j = lane; j < len; j += 64) {
        const double v = fabs(a[j] - b[j]) / S[j];
        acc = inf ? fmax(acc, v) : acc + (q == 1.0 ? v : (q == 2.0 ? v * v : pow(v, q)));
    }
    if (inf) return wave_max(acc);
    acc = wave_sum(acc);
    return q == 1.0 ? acc : (q == 2.0 ? sqrt(acc) : pow(acc, 1.0 / q));
}

template <class M>
__global__ __launch_bounds__(64) void gen_post_kernel(PostArgs a, typename M::Params par)
{
    constexpr int nx = M::nx, nu = M::nu, np = M::np, npa = np > 0 ? np : 1, ns = M::ns, nsa = ns > 0 ? ns : 1, nic = M::nic,
                  ntc = M::ntc, nbc = nic > ntc ? nic : ntc, npc = np_compact<M>(), npca = npc > 0 ? npc : 1;
    const int b = blockIdx.x, lane = threadIdx.x, N = a.N;
    if (a.active != nullptr && a.active[b] == 0) return;
    const int npt = np_total<M>(N);
    const double* pr = a.p + (long)b * npt;
    double Qu[nu], lu[nu], lx[nx], tx[nx], tp[npca], Qp[npca];
    for (int i = 0; i < npca; i++) { tp[i] = 0.0; Qp[i] = 0.0; }
    M::cost_terms(par, Qu, lu, lx, tx, tp, Qp);
    double L = 0.0, pen = 0.0, devx = 0.0, devx_tr = 0.0;
    for (int k = lane; k < N; k += 64) {
        const double* xk = a.xd + ((long)b * N + k) * nx;
        const double* uk = a.ud + ((long)b * N + k) * nu;
        const double w = trapz_w(N, k);
        double gam = 0.0;
        for (int i = 0; i < nu; i++) gam += Qu[i] * uk[i] * uk[i] + lu[i] * uk[i];
        for (int i = 0; i < nx; i++) gam += lx[i] * xk[i];
        L += w * gam;
        // terminal-cost terms of this node's own parameters (free-flyer: -eps_sdf sum(delta), definition.jl:172-184)
        for (int i = 0; i < M::np_node; i++) { const double v = pr[np + M::np_node * k + i]; L += tp[np + i] * v + Qp[np + i] * v * v; }
        double pk = 0.0;
        if (k < N - 1) for (int i = 0; i < nx; i++) pk += fabs(a.defect[((long)b * (N - 1) + k) * nx + i]);
        if (ns > 0) {
            const double tk = (1.0 - (double)k / (double)(N - 1)) * 0.0 + ((double)k / (double)(N - 1)) * 1.0;
            double s[nsa], C[nsa * nx], Dm[nsa * nu], G[nsa * npca];
            M::s_eval(par, tk, k + 1, xk, uk, pr, s, C, Dm, G);
            for (int i = 0; i < ns; i++) pk += fmax(s[i], 0.0);
        }
        pen += w * pk;
        double dxs[nx];
        for (int i = 0; i < nx; i++) dxs[i] = fabs(xk[i] - a.rxd[((long)b * N + k) * nx + i]) / a.Sx[i];
        devx = fmax(devx, qnorm_small<nx>(dxs, a.q_exit));
        devx_tr = fmax(devx_tr, qnorm_small<nx>(dxs, a.q_tr));
    }
    // ||dp||_q over the whole parameter vector, scaled: accumulate max |.|, sum |.|, sum |.|^2 and sum |.|^q per norm
    double ep = qnorm_wave(pr, a.rp + (long)b * npt, a.Sp, npt, a.q_exit, lane);
    double ep_tr = a.q_tr == a.q_exit ? ep : qnorm_wave(pr, a.rp + (long)b * npt, a.Sp, npt, a.q_tr, lane);
    L = wave_sum(L); pen = wave_sum(pen); devx = wave_max(devx); devx_tr = wave_max(devx_tr);
    if (lane == 0) {
        const double* xN = a.xd + ((long)b * N + (N - 1)) * nx;
        for (int i = 0; i < nx; i++) L += tx[i] * xN[i];
        for (int j = 0; j < np; j++) L += tp[j] * pr[j] + Qp[j] * pr[j] * pr[j];
        const double* pp = a.pp + (long)b * M::npp;
        double g[nbc], H[nbc * nx], K[nbc * npa];
        M::bc_ic(par, a.xd + (long)b * N * nx, pr, pp, g, H, K);
        for (int i = 0; i < nic; i++) pen += fabs(g[i]);
        M::bc_tc(par, xN, pr, pp, g, H, K);
        for (int i = 0; i < ntc; i++) pen += fabs(g[i]);
        a.post[(long)b * 4 + 0] = L; a.post[(long)b * 4 + 1] = pen; a.post[(long)b * 4 + 2] = ep + devx;
        // trust_region_cost(:nonconvex), gusto.jl:1172-1185: max_k (||dx_k||_q^w + ||dp||_q^w) with w = 2 for q = 4, else 1
        a.post[(long)b * 4 + 3] = a.q_tr == 4.0 ? ep_tr * ep_tr + devx_tr * devx_tr : ep_tr + devx_tr;
    }
}

}  // namespace scp

// -------------------------------------------------------------------------------------------------------------------
struct DevMap {   // affine map resident on the device
    int len = 0, nterms = 0;
    double* val0 = nullptr;
    int* ptr = nullptr;
    int* sidx = nullptr;
    double* coef = nullptr;
};

struct scp_sub {
    scp_problem* h = nullptr;
    scp::conic::Engine eng;
    scp::SrcLayout lay{};
    int nscal = 0, nsrc = 0, nfun = 0;
    DevMap maps[6];          // c, b, h, Gx, Ax, Px
    unsigned shared_mask = 0;   // arrays without any source term: one shared copy
    DevMap fun;              // linear functionals of the conic solution
    double* src = nullptr;   // interleaved [nsrc][BS]
    double* funv = nullptr;  // interleaved [nfun][BS]
    double* stage = nullptr; // host-layout staging [max(n, nfun, nscal, 8), cap]
    long stage_len = 0;
    int *ix = nullptr, *iu = nullptr, *ip = nullptr;
    double* d_pp = nullptr;
    double* post = nullptr;  // [cap][4]
    std::vector<void*> allocs;
    // outer-loop run state (SCvx: scvx.jl:459-540; GuSTO: gusto.jl:425-502)
    scp_scvx_params sp{};
    scp_gusto_params gp{};
    bool scvx_ready = false, gusto_ready = false, ptr_ready = false;
    scp_ptr_generic_params pp_{};
    double q_exit = std::numeric_limits<double>::infinity(), q_tr = std::numeric_limits<double>::infinity();   // norms of sub_post
    int B = 0, iter = 0, iter_max = 0, hist_cap = 0;
    double* post2 = nullptr; // [cap][4] GuSTO: state penalty / lambda, dynamics error, its normalisation, max s
    double *J_ref = nullptr, *hist = nullptr;   // [cap], [iter_max][cap][SCP_SCVX_HIST_WIDTH]
    int *active = nullptr, *status = nullptr, *iters_done = nullptr, *n_active = nullptr;
    std::string err;
};

#define SUB_TRY(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            s->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return SCP_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

template <class T>
static int sub_alloc(scp_sub* s, T** ptr, size_t n)
{
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) { s->err = "hipMalloc"; return SCP_ERR_ALLOC; }
    s->allocs.push_back(d);
    *ptr = (T*)d;
    return SCP_OK;
}
template <class T>
static int sub_upload(scp_sub* s, T** ptr, const T* src, size_t n)
{
    int rc = sub_alloc(s, ptr, n);
    if (rc != SCP_OK) return rc;
    if (n > 0) SUB_TRY(hipMemcpy(*ptr, src, n * sizeof(T), hipMemcpyHostToDevice));
    return SCP_OK;
}
static int sub_upload_map(scp_sub* s, DevMap& d, const scp_affine_map& m, int nsrc_limit)
{
    if (m.len < 0 || (m.len > 0 && (!m.val0 || !m.ptr))) { s->err = "bad affine map"; return SCP_ERR_BAD_ARGUMENT; }
    d.len = m.len;
    d.nterms = m.len > 0 ? m.ptr[m.len] : 0;
    if (d.nterms > 0 && (!m.sidx || !m.coef)) { s->err = "bad affine map"; return SCP_ERR_BAD_ARGUMENT; }
    for (int t = 0; t < d.nterms; t++) if (m.sidx[t] < 0 || m.sidx[t] >= nsrc_limit) { s->err = "affine map: source index out of range"; return SCP_ERR_BAD_ARGUMENT; }
    int rc;
    std::vector<int> zero_ptr(1, 0);
    if ((rc = sub_upload(s, &d.val0, m.val0, (size_t)m.len)) != SCP_OK) return rc;
    if ((rc = sub_upload(s, &d.ptr, m.len > 0 ? m.ptr : zero_ptr.data(), (size_t)m.len + 1)) != SCP_OK) return rc;
    if ((rc = sub_upload(s, &d.sidx, m.sidx, (size_t)d.nterms)) != SCP_OK) return rc;
    if ((rc = sub_upload(s, &d.coef, m.coef, (size_t)d.nterms)) != SCP_OK) return rc;
    return SCP_OK;
}

extern "C" int scp_sub_source_layout(scp_handle h, int nscal, int* offsets, int* nsrc)
{
    if (!h || nscal < 0) return SCP_ERR_BAD_ARGUMENT;
    const scp::SrcLayout L = scp::src_layout(h->info, h->N, nscal);
    if (offsets) for (int i = 0; i <= scp::SEG_COUNT; i++) offsets[i] = (int)L.off[i];
    if (nsrc) *nsrc = (int)L.off[scp::SEG_COUNT];
    return SCP_OK;
}

extern "C" const char* scp_sub_last_error(scp_sub_handle s) { return s ? s->err.c_str() : "null handle"; }

// statistics of the subproblem's conic engine: the layout of scp_conic_stats (include/scp_conic.h)
extern "C" int scp_sub_stats(scp_sub_handle s, long long stats[16])
{
    if (!s || !stats) return SCP_ERR_BAD_ARGUMENT;
    const scp::conic::Engine& E = s->eng;
    stats[0] = E.sched.nnzL; stats[1] = E.sym.flops; stats[2] = E.sched.nk; stats[3] = E.sched.nnzGt; stats[4] = E.bytes_per_problem;
    stats[5] = E.sched.nlev; stats[6] = E.sched.nrlev; stats[7] = E.waves_per_group; stats[8] = E.sym.nd_depth;
    stats[9] = E.n_fallback; stats[10] = E.n_launched; stats[11] = E.has_fb ? E.sched_fb.nlev : 0;
    stats[12] = E.n_rescued; stats[13] = stats[14] = stats[15] = 0;
    return SCP_OK;
}

extern "C" int scp_sub_destroy(scp_sub_handle s)
{
    if (!s) return SCP_ERR_BAD_ARGUMENT;
    (void)hipSetDevice(s->h->device);
    (void)hipStreamSynchronize(s->h->stream);
    s->eng.destroy();
    for (void* p : s->allocs) (void)hipFree(p);
    delete s;
    return SCP_OK;
}

extern "C" int scp_sub_create(scp_handle h, const scp_sub_template* T, scp_sub_handle* out)
{
    if (!h || !T || !out) return SCP_ERR_BAD_ARGUMENT;
    *out = nullptr;
    if (T->n < 1 || T->nscal < 0 || T->nfun < 0 || !T->ix || !T->iu || (h->npt > 0 && !T->ip)) return SCP_ERR_BAD_ARGUMENT;
    if (!h->info.has_subproblem) { h->err = "this model has no subproblem definition (discretize! / propagate / guess only)"; return SCP_ERR_UNSUPPORTED; }
    scp_sub* s = new (std::nothrow) scp_sub;
    if (!s) return SCP_ERR_ALLOC;
    s->h = h;
    auto fail = [&](int rc) { h->err = s->err; s->eng.destroy(); for (void* p : s->allocs) (void)hipFree(p); delete s; return rc; };
    if (hipSetDevice(h->device) != hipSuccess) return fail(SCP_ERR_HIP);
    s->nscal = T->nscal; s->nfun = T->nfun;
    s->lay = scp::src_layout(h->info, h->N, T->nscal);
    s->nsrc = (int)s->lay.off[scp::SEG_COUNT];
    if (T->nsrc != s->nsrc) { s->err = "template source length does not match scp_sub_source_layout"; return fail(SCP_ERR_BAD_ARGUMENT); }
    auto csc = [](int nrow, int ncol, const int* p, const int* i) {
        scp::conic::Csc M;
        M.nrow = nrow; M.ncol = ncol;
        if (!p) { M.p.assign(ncol + 1, 0); return M; }
        M.p.assign(p, p + ncol + 1);
        if (M.p[ncol] > 0 && i) M.i.assign(i, i + M.p[ncol]);
        return M;
    };
    std::vector<int> q(T->q, T->q + T->ncones);
    int rc = s->eng.create(T->n, T->p, T->m, T->l, q, csc(T->n, T->n, T->Pp, T->Pi), csc(T->p, T->n, T->Ap, T->Ai),
                           csc(T->m, T->n, T->Gp, T->Gi), nullptr, h->cap, h->device);
    if (rc != SCP_OK) { s->err = s->eng.err; return fail(rc); }
    const scp::conic::Sched& D = s->eng.sched;
    const scp_affine_map* src_maps[6] = {&T->c, &T->b, &T->h, &T->Gx, &T->Ax, &T->Px};
    const int want[6] = {D.n, D.p, D.m, D.nnzG, D.nnzA, D.nnzP};
    double* shared_dst[6] = {s->eng.c_sh, s->eng.b_sh, s->eng.h_sh, s->eng.Gx_sh, s->eng.Ax_sh, s->eng.Px_sh};
    for (int k = 0; k < 6; k++) {
        if (src_maps[k]->len != want[k]) { s->err = "affine map length does not match the conic pattern"; return fail(SCP_ERR_BAD_ARGUMENT); }
        if ((rc = sub_upload_map(s, s->maps[k], *src_maps[k], s->nsrc)) != SCP_OK) return fail(rc);
        if (s->maps[k].nterms == 0) {       // constant array: one shared copy, never gathered
            s->shared_mask |= 1u << k;
            if (want[k] > 0 && hipMemcpy(shared_dst[k], src_maps[k]->val0, sizeof(double) * want[k], hipMemcpyHostToDevice) != hipSuccess)
                return fail(SCP_ERR_HIP);
        }
    }
    if (T->nfun > 0) { if ((rc = sub_upload_map(s, s->fun, T->fun, T->n)) != SCP_OK) return fail(rc); }
    const size_t nx = h->info.nx, nu = h->info.nu, np = h->npt, N = h->N, BS = s->eng.BS, cap = h->cap;
    for (size_t e = 0; e < nx * N; e++) if (T->ix[e] < 0 || T->ix[e] >= T->n) { s->err = "ix out of range"; return fail(SCP_ERR_BAD_ARGUMENT); }
    for (size_t e = 0; e < nu * N; e++) if (T->iu[e] < 0 || T->iu[e] >= T->n) { s->err = "iu out of range"; return fail(SCP_ERR_BAD_ARGUMENT); }
    for (size_t e = 0; e < np; e++) if (T->ip[e] < 0 || T->ip[e] >= T->n) { s->err = "ip out of range"; return fail(SCP_ERR_BAD_ARGUMENT); }
    if ((rc = sub_upload(s, &s->ix, T->ix, nx * N)) != SCP_OK) return fail(rc);
    if ((rc = sub_upload(s, &s->iu, T->iu, nu * N)) != SCP_OK) return fail(rc);
    if ((rc = sub_upload(s, &s->ip, T->ip, np)) != SCP_OK) return fail(rc);
    if ((rc = sub_alloc(s, &s->src, (size_t)s->nsrc * BS)) != SCP_OK) return fail(rc);
    if ((rc = sub_alloc(s, &s->funv, (size_t)std::max(T->nfun, 1) * BS)) != SCP_OK) return fail(rc);
    s->stage_len = (long)std::max<size_t>(std::max<size_t>((size_t)T->n, (size_t)T->nfun), std::max<size_t>((size_t)T->nscal, 8)) * (long)cap;
    if ((rc = sub_alloc(s, &s->stage, (size_t)s->stage_len)) != SCP_OK) return fail(rc);
    if ((rc = sub_alloc(s, &s->d_pp, (size_t)std::max(h->info.npp, 1) * cap)) != SCP_OK) return fail(rc);
    if ((rc = sub_alloc(s, &s->post, 4 * cap)) != SCP_OK) return fail(rc);
    if (hipMemset(s->src, 0, sizeof(double) * (size_t)s->nsrc * BS) != hipSuccess) return fail(SCP_ERR_HIP);
    *out = s;
    return SCP_OK;
}

// fill the source vector from the resident reference (h->ref_*), its discretisation and the linearisations
static int sub_fill_sources(scp_sub* s, int B, const int* active)
{
    scp_problem* h = s->h;
    const long BS = s->eng.BS;
    const scp::SrcLayout& L = s->lay;
    auto tr = [&](const double* src, int seg) -> int {
        const long len = L.off[seg + 1] - L.off[seg];
        return scp::conic::transpose_to_interleaved(h->stream, src, s->src + L.off[seg] * BS, len, B, (int)BS);
    };
    int rc;
    if ((rc = tr(h->ref_xd, scp::SEG_XREF)) != SCP_OK) return rc;
    if ((rc = tr(h->ref_ud, scp::SEG_UREF)) != SCP_OK) return rc;
    if ((rc = tr(h->ref_p, scp::SEG_PREF)) != SCP_OK) return rc;
    if ((rc = tr(h->ref_dyn.A, scp::SEG_A)) != SCP_OK) return rc;
    if ((rc = tr(h->ref_dyn.Bm, scp::SEG_BM)) != SCP_OK) return rc;
    if ((rc = tr(h->ref_dyn.Bp, scp::SEG_BP)) != SCP_OK) return rc;
    if ((rc = tr(h->ref_dyn.F, scp::SEG_F)) != SCP_OK) return rc;
    if ((rc = tr(h->ref_dyn.r, scp::SEG_R)) != SCP_OK) return rc;
    if ((rc = tr(h->ref_dyn.E, scp::SEG_E)) != SCP_OK) return rc;
    scp::GenLinArgs a;
    a.B = B; a.N = h->N; a.BS = BS; a.xd = h->ref_xd; a.ud = h->ref_ud; a.p = h->ref_p; a.pp = s->d_pp; a.src = s->src;
    a.oC = L.off[scp::SEG_C]; a.oD = L.off[scp::SEG_D]; a.oG = L.off[scp::SEG_GS]; a.oRS = L.off[scp::SEG_RS];
    a.oH0 = L.off[scp::SEG_H0]; a.oK0 = L.off[scp::SEG_K0]; a.oL0 = L.off[scp::SEG_L0];
    a.oHF = L.off[scp::SEG_HF]; a.oKF = L.off[scp::SEG_KF]; a.oLF = L.off[scp::SEG_LF];
    a.active = active;
    rc = with_model(h->model_id, [&](auto m) -> int {
        using M = decltype(m);
        typename M::Params P = M::make_params(h->par.data());
        const long total = (long)B * (h->N + 1);
        hipLaunchKernelGGL(scp::gen_linearise_kernel<M>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, a, P);
        return (int)SCP_OK;
    });
    if (rc != SCP_OK) return rc;
    SUB_TRY(hipGetLastError());
    return SCP_OK;
}

// gather the conic values, solve, read x / u / p out into h->sol_*, discretise the new point
static int sub_solve_dev(scp_sub* s, int B, const scp::conic::Opts& o, const int* active)
{
    scp_problem* h = s->h;
    scp::conic::Engine& E = s->eng;
    const long BS = E.BS;
    double* dst[6] = {E.c, E.b, E.h, E.Gx, E.Ax, E.Px};
    for (int k = 0; k < 6; k++) {
        if (s->shared_mask & (1u << k)) continue;
        const DevMap& m = s->maps[k];
        if (m.len == 0) continue;
        scp::GatherArgs g;
        g.B = B; g.len = m.len; g.BS = BS; g.val0 = m.val0; g.ptr = m.ptr; g.sidx = m.sidx; g.coef = m.coef; g.src = s->src;
        g.dst = dst[k];
        hipLaunchKernelGGL(scp::gen_gather_kernel, dim3((B + 63) / 64, (m.len + 3) / 4), dim3(256), 0, h->stream, g);
    }
    SUB_TRY(hipGetLastError());
    TRY(stamp_begin(h, 2));
    int rc = E.launch(h->stream, B, o, s->shared_mask, active);
    if (rc != SCP_OK) { s->err = E.err; return rc; }
    TRY(stamp_end(h));
    const int nx = h->info.nx, nu = h->info.nu, np = h->npt, N = h->N;
    auto ro = [&](const int* idx, int len, int dim, const double* S, const double* c, double* out) {
        if (len == 0) return;
        scp::ReadoutArgs r;
        r.B = B; r.len = len; r.dim = dim; r.BS = BS; r.idx = idx; r.S = S; r.c = c; r.x = E.x; r.out = out; r.active = active;
        hipLaunchKernelGGL(scp::gen_readout_kernel, dim3((B + 63) / 64, (len + 3) / 4), dim3(256), 0, h->stream, r);
    };
    ro(s->ix, nx * N, nx, h->d_Sx, h->d_cx, h->sol_xd);
    ro(s->iu, nu * N, nu, h->d_Su, h->d_cu, h->sol_ud);
    ro(s->ip, np, np > 0 ? np : 1, h->d_Sp, h->d_cp, h->sol_p);
    if (s->nfun > 0) {
        scp::GatherArgs g;
        g.B = B; g.len = s->nfun; g.BS = BS; g.val0 = s->fun.val0; g.ptr = s->fun.ptr; g.sidx = s->fun.sidx; g.coef = s->fun.coef;
        g.src = E.x; g.dst = s->funv;
        hipLaunchKernelGGL(scp::gen_gather_kernel, dim3((B + 63) / 64, (s->nfun + 3) / 4), dim3(256), 0, h->stream, g);
    }
    SUB_TRY(hipGetLastError());
    TRY(discretize_dev(h, B, h->sol_xd, h->sol_ud, h->sol_p, h->sol_dyn, h->d_feas_new, active));
    // d_feas keeps the flag of every problem's LAST solution (d_feas_new is only meaningful for the problems of this launch)
    hipLaunchKernelGGL(merge_feas_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, B, active, h->d_feas_new, h->d_feas);
    SUB_TRY(hipGetLastError());
    return SCP_OK;
}

static scp::conic::Opts sub_opts(const scp_conic_opts* opts)
{
    scp::conic::Opts o = scp::conic::default_opts();
    if (opts) {
        o.max_iter = opts->max_iter; o.feastol = opts->feastol; o.abstol = opts->abstol; o.reltol = opts->reltol;
        o.reg = opts->reg; o.dyn_eps = opts->dyn_eps; o.dyn_delta = opts->dyn_delta; o.nref = opts->nref;
        o.ref_tol = opts->ref_tol; o.step = opts->step;
    }
    return o;
}

static int sub_put_scal(scp_sub* s, int B, const double* scal)
{
    if (s->nscal == 0) return SCP_OK;
    scp_problem* h = s->h;
    SUB_TRY(hipMemcpyAsync(s->stage, scal, sizeof(double) * s->nscal * B, hipMemcpyHostToDevice, h->stream));
    return scp::conic::transpose_to_interleaved(h->stream, s->stage, s->src + s->lay.off[scp::SEG_SCAL] * s->eng.BS, s->nscal, B,
                                                s->eng.BS);
}
static int sub_get_il(scp_sub* s, int B, const double* src_il, long len, double* dst)
{
    if (!dst || len == 0) return SCP_OK;
    scp_problem* h = s->h;
    int rc = scp::conic::transpose_from_interleaved(h->stream, src_il, s->stage, len, B, s->eng.BS);
    if (rc != SCP_OK) return rc;
    SUB_TRY(hipMemcpyAsync(dst, s->stage, sizeof(double) * len * B, hipMemcpyDeviceToHost, h->stream));
    SUB_TRY(hipStreamSynchronize(h->stream));
    return SCP_OK;
}

extern "C" int scp_sub_solve_batch_host(scp_sub_handle s, int B, const double* xd_ref, const double* ud_ref, const double* p_ref,
                                        const double* pp, const double* scal, const scp_conic_opts* opts, double* x, double* u,
                                        double* p, double* fun, double* xconic, int32_t* status, int32_t* iters, double* info,
                                        double* defect, uint8_t* feas, double* seconds)
{
    if (!s || B < 1 || !xd_ref || !ud_ref) return SCP_ERR_BAD_ARGUMENT;
    scp_problem* h = s->h;
    if (B > h->cap) { s->err = "batch size exceeds batch_capacity"; return SCP_ERR_BATCH_TOO_LARGE; }
    if ((h->npt > 0 && !p_ref) || (h->info.npp > 0 && !pp) || (s->nscal > 0 && !scal)) { s->err = "missing input"; return SCP_ERR_BAD_ARGUMENT; }
    SUB_TRY(hipSetDevice(h->device));
    TRY(upload_traj(h, B, xd_ref, ud_ref, p_ref, h->ref_xd, h->ref_ud, h->ref_p));
    if (h->info.npp > 0) SUB_TRY(hipMemcpyAsync(s->d_pp, pp, sizeof(double) * h->info.npp * B, hipMemcpyHostToDevice, h->stream));
    SUB_TRY(hipEventRecord(h->ev0, h->stream));
    TRY(discretize_dev(h, B, h->ref_xd, h->ref_ud, h->ref_p, h->ref_dyn, h->d_feas, nullptr));
    int rc;
    if ((rc = sub_fill_sources(s, B, nullptr)) != SCP_OK) return rc;
    if ((rc = sub_put_scal(s, B, scal)) != SCP_OK) return rc;
    if ((rc = sub_solve_dev(s, B, sub_opts(opts), nullptr)) != SCP_OK) return rc;
    SUB_TRY(hipEventRecord(h->ev1, h->stream));
    const size_t nx = h->info.nx, nu = h->info.nu, np = h->npt, N = h->N, D = sizeof(double), b = B;
    if (x) SUB_TRY(hipMemcpyAsync(x, h->sol_xd, nx * N * b * D, hipMemcpyDeviceToHost, h->stream));
    if (u) SUB_TRY(hipMemcpyAsync(u, h->sol_ud, nu * N * b * D, hipMemcpyDeviceToHost, h->stream));
    if (p && np > 0) SUB_TRY(hipMemcpyAsync(p, h->sol_p, np * b * D, hipMemcpyDeviceToHost, h->stream));
    if (defect) SUB_TRY(hipMemcpyAsync(defect, h->sol_dyn.defect, nx * (N - 1) * b * D, hipMemcpyDeviceToHost, h->stream));
    if (status) SUB_TRY(hipMemcpyAsync(status, s->eng.status, sizeof(int) * b, hipMemcpyDeviceToHost, h->stream));
    if (iters) SUB_TRY(hipMemcpyAsync(iters, s->eng.iters, sizeof(int) * b, hipMemcpyDeviceToHost, h->stream));
    TRY(feas_out(h, B, h->d_feas_new, feas));
    if ((rc = sub_get_il(s, B, s->funv, s->nfun, fun)) != SCP_OK) return rc;
    if ((rc = sub_get_il(s, B, s->eng.x, s->eng.sched.n, xconic)) != SCP_OK) return rc;
    if ((rc = sub_get_il(s, B, s->eng.info, 8, info)) != SCP_OK) return rc;
    if (seconds) { float ms = 0; SUB_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1)); *seconds = ms * 1e-3; }
    return SCP_OK;
}

// -------------------------------------------------------------------------------------------------------------------
// SCvx outer loop on the device (src/solvers/scvx.jl:459-540)
// -------------------------------------------------------------------------------------------------------------------
namespace scp {

enum { SH_L = 0, SH_LPEN, SH_LAUG, SH_JREF, SH_JSOL, SH_PRE, SH_ACT, SH_RHO, SH_ETA, SH_ETANEXT, SH_ACCEPT, SH_STOP, SH_DEV,
       SH_FEAS, SH_STATUS, SH_IPMIT, SH_N = SCP_SCVX_HIST_WIDTH };

struct ScvxUpdateArgs {
    int B, iter;
    long BS;
    scp_scvx_params sp;
    const double* post;      // [B][4]: L, nonlinear penalty / lambda, deviation
    const double* fun;       // interleaved [nfun][BS]: fun[0] = trapz(P) + sum(Pf) of the subproblem
    const int* feas;         // [B] of the new point
    const int* ipm_status;
    const int* ipm_iters;
    double* J_ref;           // [B] nonlinear cost of the reference (in/out)
    double* J_last;          // [B] nonlinear cost of the last solution
    double* eta;             // interleaved scal segment of the source vector: eta[b]
    int* active;
    int* accept;             // [B] out: ref <- sol
    int* scp_status;
    int* iters_done;
    double* hist;            // [iter_max][B][SH_N]
    int* n_active;
};

// check_stopping_criterion! (scvx.jl:711-734) + update_trust_region! (:753-769) + update_rule (:1000-1045)
__global__ void scvx_update_kernel(ScvxUpdateArgs a)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    a.accept[b] = 0;
    if (!a.active[b]) return;
    double* h = a.hist + ((long)(a.iter - 1) * a.B + b) * SH_N;
    const scp_scvx_params& sp = a.sp;
    const double L = a.post[(long)b * 4 + 0], pen = a.post[(long)b * 4 + 1], dev = a.post[(long)b * 4 + 2];
    const double Lpen = sp.lam * a.fun[b];                       // scvx.jl:895-898
    const double eta = a.eta[b];
    const bool unsafe = a.ipm_status[b] > 1;                     // unsafe_solution, scp.jl:965-980
    const bool feas = a.feas[b] != 0;
    const double J_ref = a.J_ref[b];
    const double pre = J_ref - L;                                // predicted improvement uses the ORIGINAL cost (scvx.jl:972-973)
    const double pre_rel = pre / fabs(J_ref);
    const bool stop = a.iter > 1 && (feas && (pre_rel <= sp.eps_rel || dev <= sp.eps_abs));
    const double J_sol = L + sp.lam * pen;                       // actual_cost_penalty!, scvx.jl:924-952
    const double act = J_ref - J_sol;
    const double rho = act / pre;
    h[SH_L] = L; h[SH_LPEN] = Lpen; h[SH_LAUG] = L + Lpen; h[SH_JREF] = J_ref; h[SH_JSOL] = J_sol; h[SH_PRE] = pre;
    h[SH_ACT] = act; h[SH_RHO] = rho; h[SH_ETA] = eta; h[SH_DEV] = dev; h[SH_FEAS] = feas ? 1.0 : 0.0;
    h[SH_STATUS] = (double)a.ipm_status[b]; h[SH_IPMIT] = (double)a.ipm_iters[b]; h[SH_STOP] = stop ? 1.0 : 0.0;
    h[SH_ACCEPT] = 0.0; h[SH_ETANEXT] = eta;
    a.iters_done[b] = a.iter;
    a.J_last[b] = J_sol;
    if (unsafe) { a.scp_status[b] = 1; a.active[b] = 0; return; }
    if (stop) { a.active[b] = 0; return; }
    bool acc;
    double eta_next;
    if (rho < sp.rho_0) { acc = false; eta_next = fmax(sp.eta_lb, eta / sp.beta_sh); }
    else if (rho < sp.rho_1) { acc = true; eta_next = fmax(sp.eta_lb, eta / sp.beta_sh); }
    else if (rho < sp.rho_2) { acc = true; eta_next = eta; }
    else { acc = true; eta_next = fmin(sp.eta_ub, sp.beta_gr * eta); }
    if (!(rho == rho)) { acc = false; eta_next = fmax(sp.eta_lb, eta / sp.beta_sh); }   // NaN ratio: treat as rejected
    h[SH_ACCEPT] = acc ? 1.0 : 0.0; h[SH_ETANEXT] = eta_next;
    a.accept[b] = acc ? 1 : 0;
    a.eta[b] = eta_next;
    if (acc) a.J_ref[b] = J_sol;
    if (a.iter >= sp.iter_max) { a.active[b] = 0; return; }
    atomicAdd(a.n_active, 1);
}

// dst[b] <- src[b] for the problems with mask[b] != 0 (len doubles per problem, problem-major arrays)
__global__ __launch_bounds__(256) void masked_copy_kernel(double* dst, const double* src, long len, const int* mask, int B)
{
    const int b = blockIdx.y;
    if (b >= B || mask[b] == 0) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < len; i += (long)gridDim.x * 256) dst[(long)b * len + i] = src[(long)b * len + i];
}
__global__ void fill_strided_kernel(double* dst, double v, int B) { const int b = blockIdx.x * blockDim.x + threadIdx.x; if (b < B) dst[b] = v; }
__global__ void jref_kernel(const double* post, double lam, double* J, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) J[b] = post[(long)b * 4 + 0] + lam * post[(long)b * 4 + 1];
}
__global__ void proj_status_kernel(const int* ipm_status, int* active, int* scp_status, int* accept, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const bool ok = ipm_status[b] <= 1;
    accept[b] = ok ? 1 : 0;
    if (!ok) { active[b] = 0; scp_status[b] = 2; }   // SCP_GUESS_PROJECTION_FAILED (scp.jl:352-357)
}

}  // namespace scp

static int sub_masked_copy_all(scp_sub* s, int B, const int* mask, bool to_ref)
{
    scp_problem* h = s->h;
    const long nx = h->info.nx, nu = h->info.nu, np = h->npt, npF = h->info.npF, N = h->N, M = N - 1;
    auto cp = [&](double* a, double* b_, long len) {
        if (len == 0) return;
        double* dst = to_ref ? a : b_;
        const double* src = to_ref ? b_ : a;
        const unsigned gx = (unsigned)std::min<long>((len + 255) / 256, 64);
        hipLaunchKernelGGL(scp::masked_copy_kernel, dim3(gx, B), dim3(256), 0, h->stream, dst, src, len, mask, B);
    };
    cp(h->ref_xd, h->sol_xd, nx * N); cp(h->ref_ud, h->sol_ud, nu * N); cp(h->ref_p, h->sol_p, np);
    cp(h->ref_dyn.A, h->sol_dyn.A, nx * nx * M); cp(h->ref_dyn.Bm, h->sol_dyn.Bm, nx * nu * M); cp(h->ref_dyn.Bp, h->sol_dyn.Bp, nx * nu * M);
    cp(h->ref_dyn.F, h->sol_dyn.F, nx * npF * M); cp(h->ref_dyn.r, h->sol_dyn.r, nx * M); cp(h->ref_dyn.E, h->sol_dyn.E, nx * nx * M);
    cp(h->ref_dyn.defect, h->sol_dyn.defect, nx * M);
    SUB_TRY(hipGetLastError());
    return SCP_OK;
}

static int sub_post(scp_sub* s, int B, const double* xd, const double* ud, const double* p, const double* defect, const int* active)
{
    scp_problem* h = s->h;
    scp::PostArgs a;
    a.q_exit = s->q_exit; a.q_tr = s->q_tr;
    a.B = B; a.N = h->N; a.xd = xd; a.ud = ud; a.p = p; a.pp = s->d_pp; a.defect = defect; a.rxd = h->ref_xd; a.rp = h->ref_p;
    a.Sx = h->d_Sx; a.Sp = h->d_Sp; a.post = s->post; a.active = active;
    int rc = with_model(h->model_id, [&](auto m) -> int {
        using M = decltype(m);
        typename M::Params P = M::make_params(h->par.data());
        hipLaunchKernelGGL(scp::gen_post_kernel<M>, dim3(B), dim3(64), 0, h->stream, a, P);
        return (int)SCP_OK;
    });
    if (rc != SCP_OK) return rc;
    SUB_TRY(hipGetLastError());
    return SCP_OK;
}

// run state shared by the outer loops: J_ref/J_last [2 cap], hist [iter_max][cap][16], active/status/iters/n_active/accept
static int sub_loop_state(scp_sub* s, int iter_max)
{
    scp_problem* h = s->h;
    int rc;
    if (s->J_ref == nullptr) {
        if ((rc = sub_alloc(s, &s->J_ref, 2 * (size_t)h->cap)) != SCP_OK) return rc;
        if ((rc = sub_alloc(s, &s->active, 4 * (size_t)h->cap + 1)) != SCP_OK) return rc;
        if ((rc = sub_alloc(s, &s->post2, 4 * (size_t)h->cap)) != SCP_OK) return rc;
        s->status = s->active + h->cap; s->iters_done = s->status + h->cap; s->n_active = s->iters_done + h->cap;
    }
    if (s->hist_cap < iter_max) {
        if ((rc = sub_alloc(s, &s->hist, (size_t)iter_max * h->cap * SCP_SCVX_HIST_WIDTH)) != SCP_OK) return rc;
        s->hist_cap = iter_max;
    }
    return SCP_OK;
}

extern "C" int scp_scvx_init_host(scp_sub_handle s, scp_sub_handle proj, int B, const scp_scvx_params* pars, const double* xd,
                                  const double* ud, const double* p, const double* pp)
{
    if (!s || !pars || B < 1 || !xd || !ud) return SCP_ERR_BAD_ARGUMENT;
    scp_problem* h = s->h;
    if (B > h->cap) { s->err = "batch size exceeds batch_capacity"; return SCP_ERR_BATCH_TOO_LARGE; }
    if ((h->npt > 0 && !p) || (h->info.npp > 0 && !pp)) { s->err = "missing input"; return SCP_ERR_BAD_ARGUMENT; }
    if (pars->iter_max < 1 || s->nscal != 1 || s->nfun < 1) { s->err = "not an SCvx template (nscal = 1: eta, fun[0] = penalty)"; return SCP_ERR_BAD_ARGUMENT; }
    if (proj && proj->h != h) { s->err = "projection template belongs to another problem handle"; return SCP_ERR_BAD_ARGUMENT; }
    SUB_TRY(hipSetDevice(h->device));
    int rc;
    if ((rc = sub_loop_state(s, pars->iter_max)) != SCP_OK) return rc;
    if (!(pars->q_exit >= 1.0)) { s->err = "q_exit must be >= 1 (or Inf)"; return SCP_ERR_BAD_ARGUMENT; }
    s->sp = *pars; s->B = B; s->iter = 0; s->iter_max = pars->iter_max; s->scvx_ready = true; s->gusto_ready = false; s->ptr_ready = false;
    s->q_exit = pars->q_exit; s->q_tr = pars->q_exit;
    TRY(upload_traj(h, B, xd, ud, p, h->ref_xd, h->ref_ud, h->ref_p));
    if (h->info.npp > 0) {
        SUB_TRY(hipMemcpyAsync(s->d_pp, pp, sizeof(double) * h->info.npp * B, hipMemcpyHostToDevice, h->stream));
        if (proj) SUB_TRY(hipMemcpyAsync(proj->d_pp, pp, sizeof(double) * h->info.npp * B, hipMemcpyHostToDevice, h->stream));
    }
    SUB_TRY(hipMemsetAsync(s->hist, 0, sizeof(double) * (size_t)pars->iter_max * B * SCP_SCVX_HIST_WIDTH, h->stream));
    SUB_TRY(hipMemsetAsync(s->status, 0, sizeof(int) * (size_t)B, h->stream));
    SUB_TRY(hipMemsetAsync(s->iters_done, 0, sizeof(int) * (size_t)B, h->stream));
    std::vector<int> ones(B, 1);
    SUB_TRY(hipMemcpyAsync(s->active, ones.data(), sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    SUB_TRY(hipStreamSynchronize(h->stream));
    // ---- generate_initial_guess (scvx.jl:555-565): correct_convex! projects the guess onto the convex sets ----
    if (proj) {
        TRY(discretize_dev(h, B, h->ref_xd, h->ref_ud, h->ref_p, h->ref_dyn, h->d_feas, nullptr));
        if ((rc = sub_fill_sources(proj, B, nullptr)) != SCP_OK) { s->err = proj->err; return rc; }
        if ((rc = sub_solve_dev(proj, B, sub_opts(&pars->solver), nullptr)) != SCP_OK) { s->err = proj->err; return rc; }
        int* acc = s->active + 3 * (size_t)h->cap + 1;
        hipLaunchKernelGGL(scp::proj_status_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, proj->eng.status, s->active,
                           s->status, acc, B);
        if ((rc = sub_masked_copy_all(s, B, acc, true)) != SCP_OK) return rc;   // x_ref .= value(opti.x) (scp.jl:346-349)
    }
    TRY(discretize_dev(h, B, h->ref_xd, h->ref_ud, h->ref_p, h->ref_dyn, h->d_feas, nullptr));
    // nonlinear cost of the initial reference (solution_cost!(ref, :nonlinear), scvx.jl:724)
    if ((rc = sub_post(s, B, h->ref_xd, h->ref_ud, h->ref_p, h->ref_dyn.defect, nullptr)) != SCP_OK) return rc;
    hipLaunchKernelGGL(scp::jref_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, s->post, pars->lam, s->J_ref, B);
    hipLaunchKernelGGL(scp::fill_strided_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream,
                       s->src + s->lay.off[scp::SEG_SCAL] * s->eng.BS, pars->eta_init, B);
    SUB_TRY(hipGetLastError());
    SUB_TRY(hipStreamSynchronize(h->stream));
    stamps_collect(h);
    return SCP_OK;
}

extern "C" int scp_scvx_iterate(scp_sub_handle s, int* n_active)
{
    if (!s || !s->scvx_ready) return SCP_ERR_BAD_ARGUMENT;
    scp_problem* h = s->h;
    SUB_TRY(hipSetDevice(h->device));
    if (s->iter >= s->sp.iter_max) { if (n_active) *n_active = 0; return SCP_OK; }
    const int B = s->B;
    s->iter += 1;
    int rc;
    int* acc = s->active + 3 * (size_t)h->cap + 1;
    SUB_TRY(hipMemsetAsync(s->n_active, 0, sizeof(int), h->stream));
    if ((rc = sub_fill_sources(s, B, s->active)) != SCP_OK) return rc;
    if ((rc = sub_solve_dev(s, B, sub_opts(&s->sp.solver), s->active)) != SCP_OK) return rc;
    if ((rc = sub_post(s, B, h->sol_xd, h->sol_ud, h->sol_p, h->sol_dyn.defect, s->active)) != SCP_OK) return rc;
    scp::ScvxUpdateArgs a;
    a.B = B; a.iter = s->iter; a.BS = s->eng.BS; a.sp = s->sp; a.post = s->post; a.fun = s->funv; a.feas = h->d_feas_new;
    a.ipm_status = s->eng.status; a.ipm_iters = s->eng.iters; a.J_ref = s->J_ref; a.J_last = s->J_ref + h->cap;
    a.eta = s->src + s->lay.off[scp::SEG_SCAL] * s->eng.BS; a.active = s->active; a.accept = acc; a.scp_status = s->status;
    a.iters_done = s->iters_done; a.hist = s->hist; a.n_active = s->n_active;
    hipLaunchKernelGGL(scp::scvx_update_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, a);
    SUB_TRY(hipGetLastError());
    if ((rc = sub_masked_copy_all(s, B, acc, true)) != SCP_OK) return rc;    // ref = sol for the accepted steps
    int na = 0;
    SUB_TRY(hipMemcpyAsync(&na, s->n_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    SUB_TRY(hipStreamSynchronize(h->stream));
    stamps_collect(h);
    if (n_active) *n_active = na;
    return SCP_OK;
}

extern "C" int scp_scvx_get_host(scp_sub_handle s, double* xd, double* ud, double* p, int32_t* status, int32_t* iterations,
                                 double* cost, uint8_t* feas, double* defect, double* hist)
{
    if (!s || !(s->scvx_ready || s->gusto_ready || s->ptr_ready)) return SCP_ERR_BAD_ARGUMENT;
    scp_problem* h = s->h;
    SUB_TRY(hipSetDevice(h->device));
    const size_t nx = h->info.nx, nu = h->info.nu, np = h->npt, N = h->N, D = sizeof(double), b = s->B;
    // SCPSolution(history): the LAST subproblem's solution (scp.jl:196-245); before the first iteration: the reference
    const bool none = s->iter == 0;
    if (xd) SUB_TRY(hipMemcpyAsync(xd, none ? h->ref_xd : h->sol_xd, nx * N * b * D, hipMemcpyDeviceToHost, h->stream));
    if (ud) SUB_TRY(hipMemcpyAsync(ud, none ? h->ref_ud : h->sol_ud, nu * N * b * D, hipMemcpyDeviceToHost, h->stream));
    if (p && np > 0) SUB_TRY(hipMemcpyAsync(p, none ? h->ref_p : h->sol_p, np * b * D, hipMemcpyDeviceToHost, h->stream));
    if (defect) SUB_TRY(hipMemcpyAsync(defect, none ? h->ref_dyn.defect : h->sol_dyn.defect, nx * (N - 1) * b * D, hipMemcpyDeviceToHost, h->stream));
    if (status) SUB_TRY(hipMemcpyAsync(status, s->status, sizeof(int) * b, hipMemcpyDeviceToHost, h->stream));
    if (iterations) SUB_TRY(hipMemcpyAsync(iterations, s->iters_done, sizeof(int) * b, hipMemcpyDeviceToHost, h->stream));
    if (cost) {
        SUB_TRY(hipMemcpyAsync(cost, s->J_ref, D * b, hipMemcpyDeviceToHost, h->stream));                  // J of the reference
        SUB_TRY(hipMemcpyAsync(cost + b, s->J_ref + h->cap, D * b, hipMemcpyDeviceToHost, h->stream));     // J of the last solution
    }
    if (hist) SUB_TRY(hipMemcpyAsync(hist, s->hist, D * (size_t)s->iter_max * b * SCP_SCVX_HIST_WIDTH, hipMemcpyDeviceToHost, h->stream));
    TRY(feas_out(h, s->B, h->d_feas, feas));
    return SCP_OK;
}

// -------------------------------------------------------------------------------------------------------------------
// GuSTO outer loop on the device (src/solvers/gusto.jl:425-502), quadratic penalty
// -------------------------------------------------------------------------------------------------------------------
namespace scp {

enum { GH_L = 0, GH_LST, GH_LTR, GH_JAUG, GH_JST, GH_RHO, GH_ETA, GH_LAM, GH_ETANEXT, GH_LAMNEXT, GH_FLAGS, GH_DEV, GH_STATUS,
       GH_IPMIT, GH_DYNERR, GH_DYNNRML };

// Per-problem quantities of the new point GuSTO's update needs (wave per problem, lanes over the nodes):
//   post2[0] = trapz_k (sum_i max(q_i(x_k, p), 0)^2 + sum_i max(s_i(x_k, p), 0)^2)   state_penalty_cost(:nonconvex) / lambda,
//              gusto.jl:835-865: q = cone indicators of the convex state set X, s = non-convex constraints
//   post2[1] = trapz_k ||f(x_k,u_k,p) - f_lin,k||_2          dynamics error, gusto.jl:1269-1287
//   post2[2] = trapz_k ||f_lin,k||_2                         its normalisation
//   post2[3] = max_k,i (q_i, s_i)(x_k, p)                    feasibility of the new point, gusto.jl:1342-1362
// f_lin,k = f(ref_k) + A (x - x_ref) + B (u - u_ref) + F (p - p_ref) with the Jacobians at the reference node.
struct GustoPostArgs {
    int B, N;
    const double *xd, *ud, *p;      // new point
    const double *rxd, *rud, *rp;   // reference
    double* post2;
    const int* active;
    int pen;          // 0 :quad, 1 :softplus (numerical mode of soft_penalty, gusto.jl:966-1000)
    double hom;
};
// lambda-free penalty of one quantity f: max(0, f)^2 or logsumexp([0, f]; t = hom) (src/utils/helper.jl:623-640, stable form)
__device__ __forceinline__ double gusto_pen(double f, int pen, double hom)
{
    if (pen == 0) { const double v = fmax(f, 0.0); return v * v; }
    const double a = fmax(0.0, hom * f);
    return (a + log(exp(0.0 - a) + exp(hom * f - a))) / hom;
}
template <class M>
__global__ __launch_bounds__(64) void gusto_post_kernel(GustoPostArgs a, typename M::Params par)
{
    constexpr int nx = M::nx, nu = M::nu, npF = M::npF, npFa = npF > 0 ? npF : 1, ns = M::ns, nsa = ns > 0 ? ns : 1,
                  npc = np_compact<M>(), npca = npc > 0 ? npc : 1;
    const int b = blockIdx.x, lane = threadIdx.x, N = a.N;
    if (a.active != nullptr && a.active[b] == 0) return;
    const double* pn = a.p + (long)b * np_total<M>(N);
    const double* pr = a.rp + (long)b * np_total<M>(N);
    double pen = 0.0, de = 0.0, dn = 0.0, smax = -1e300;
    for (int k = lane; k < N; k += 64) {
        const double w = trapz_w(N, k);
        const double tk = (1.0 - (double)k / (double)(N - 1)) * 0.0 + ((double)k / (double)(N - 1)) * 1.0;
        double x[nx], u[nu], xr[nx], ur[nu];
        for (int i = 0; i < nx; i++) { x[i] = a.xd[((long)b * N + k) * nx + i]; xr[i] = a.rxd[((long)b * N + k) * nx + i]; }
        for (int i = 0; i < nu; i++) { u[i] = a.ud[((long)b * N + k) * nu + i]; ur[i] = a.rud[((long)b * N + k) * nu + i]; }
        double fr[nx], Am[nx * nx], Bm[nx * nu], Fc[nx * npFa], fn[nx], A2[nx * nx], B2[nx * nu], F2[nx * npFa];
        M::dyn(par, tk, k + 1, xr, ur, pr, fr, Am, Bm, Fc);
        M::dyn(par, tk, k + 1, x, u, pn, fn, A2, B2, F2);
        double e2 = 0.0, n2 = 0.0;
        for (int i = 0; i < nx; i++) {
            double fl = fr[i];
            for (int j = 0; j < nx; j++) fl += Am[i + nx * j] * (x[j] - xr[j]);
            for (int j = 0; j < nu; j++) fl += Bm[i + nx * j] * (u[j] - ur[j]);
            for (int j = 0; j < npF; j++) fl += Fc[i + nx * j] * (pn[M::Fcol(j)] - pr[M::Fcol(j)]);
            e2 += (fn[i] - fl) * (fn[i] - fl); n2 += fl * fl;
        }
        de += w * sqrt(e2); dn += w * sqrt(n2);
        double pk = 0.0;
        // convex state set X through its cone indicators (convex_state_penalty, gusto.jl:835-865; feasibility :1342-1355)
        for_each_x_indicator<M>(par, tk, k + 1, x, pn, N, [&](double q) { pk += gusto_pen(q, a.pen, a.hom); smax = fmax(smax, q); });
        if (ns > 0) {
            double s[nsa], C[nsa * nx], Dm[nsa * nu], G[nsa * npca], uz[nu];
            for (int i = 0; i < nu; i++) uz[i] = 0.0;
            M::s_eval(par, tk, k + 1, x, uz, pn, s, C, Dm, G);
            for (int i = 0; i < ns; i++) { pk += gusto_pen(s[i], a.pen, a.hom); smax = fmax(smax, s[i]); }
        }
        pen += w * pk;
    }
    pen = wave_sum(pen); de = wave_sum(de); dn = wave_sum(dn); smax = wave_max(smax);
    if (lane == 0) {
        double* o = a.post2 + (long)b * 4;
        o[0] = pen; o[1] = de; o[2] = dn; o[3] = smax;
    }
}

struct GustoUpdateArgs {
    int B, iter, N, nst;
    long BS;
    scp_gusto_params gp;
    const double* post;      // [B][4]: L, -, deviation
    const double* post2;     // [B][4]
    const double* fun;       // interleaved [N + N nst][BS]: v_tr[k], then v_st[k][i]
    const int* feas;         // [B] dynamic feasibility of the new point
    const int* ipm_status;
    const int* ipm_iters;
    double* J_ref;
    double* J_last;
    double* scal;            // interleaved scal segment: eta[b], lambda[BS + b]
    int* active;
    int* accept;
    int* scp_status;
    int* iters_done;
    double* hist;
    int* n_active;
};

// check_stopping_criterion! (gusto.jl:1203-1230) + update_trust_region! (:1245-1293) + update_rule! (:1310-1427)
__global__ void gusto_update_kernel(GustoUpdateArgs a)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    a.accept[b] = 0;
    if (!a.active[b]) return;
    double* h = a.hist + ((long)(a.iter - 1) * a.B + b) * SCP_SCVX_HIST_WIDTH;
    const scp_gusto_params& gp = a.gp;
    const double eta = a.scal[b], lam = a.scal[a.BS + b];
    const double L = a.post[(long)b * 4 + 0], dev = a.post[(long)b * 4 + 2], dev_tr = a.post[(long)b * 4 + 3];
    double ltr = 0.0, lst = 0.0;
    for (int k = 0; k < a.N; k++) {
        const double w = trapz_w(a.N, k);
        // penalty variables of the template: v (cost lambda v^2, :quad) or w (cost lambda w / hom, :softplus, gusto.jl:1029)
        const double v = a.fun[(long)k * a.BS + b];
        ltr += w * (gp.pen == 0 ? v * v : v / gp.hom);
        double acc = 0.0;
        for (int i = 0; i < a.nst; i++) { const double vs = a.fun[((long)a.N + (long)k * a.nst + i) * a.BS + b]; acc += gp.pen == 0 ? vs * vs : vs / gp.hom; }
        lst += w * acc;
    }
    const double L_tr = lam * ltr, L_st = lam * lst, L_aug = L + L_st + L_tr;         // gusto.jl:534-550
    const double J_st = lam * a.post2[(long)b * 4 + 0];
    const double J_aug = L + J_st + L_tr;                                             // gusto.jl:399-402
    const double dyn_err = a.post2[(long)b * 4 + 1], dyn_nrml = a.post2[(long)b * 4 + 2], smax = a.post2[(long)b * 4 + 3];
    const bool unsafe = a.ipm_status[b] > 1;
    const bool solfeas = a.feas[b] != 0;
    const double J_ref = a.J_ref[b];
    const double dJ = fabs(J_ref - J_aug) / fabs(J_ref);
    const bool stop = a.iter > 1 && ((solfeas && (dJ <= gp.eps_rel || dev <= gp.eps_abs)) || lam > gp.lam_max);
    const double rho = (fabs(J_aug - L_aug) + dyn_err) / (fabs(L_aug) + dyn_nrml);
    h[GH_L] = L; h[GH_LST] = L_st; h[GH_LTR] = L_tr; h[GH_JAUG] = J_aug; h[GH_JST] = J_st; h[GH_RHO] = rho; h[GH_ETA] = eta;
    h[GH_LAM] = lam; h[GH_ETANEXT] = eta; h[GH_LAMNEXT] = lam; h[GH_DEV] = dev; h[GH_STATUS] = (double)a.ipm_status[b];
    h[GH_IPMIT] = (double)a.ipm_iters[b]; h[GH_DYNERR] = dyn_err; h[GH_DYNNRML] = dyn_nrml;
    int flags = (stop ? 2 : 0) | (solfeas ? 16 : 0);
    h[GH_FLAGS] = (double)flags;
    a.iters_done[b] = a.iter;
    a.J_last[b] = J_aug;
    if (unsafe) { a.scp_status[b] = 1; a.active[b] = 0; return; }
    if (stop) { a.active[b] = 0; return; }
    // trust_region_cost(:nonconvex) per node is ||dx_k||_q + ||dp||_q - eta in the TRUST-REGION norm q_tr; its max over k is
    // (the q_tr deviation) - eta (:1172-1185)
    const bool trust_viol = dev_tr - eta > 1e-3;
    const bool feasible = !(smax > 1e-3);
    bool acc;
    double eta_n, lam_n;
    if (trust_viol) { acc = false; eta_n = eta; lam_n = gp.gamma_fail * lam; }
    else if (rho < gp.rho_1) {
        acc = true;
        eta_n = rho < gp.rho_0 ? fmin(gp.eta_ub, gp.beta_gr * eta) : eta;
        lam_n = feasible ? gp.lam_init : gp.gamma_fail * lam;
    } else { acc = false; eta_n = fmax(gp.eta_lb, eta / gp.beta_sh); lam_n = lam; }
    if (a.iter >= gp.iter_mu) eta_n *= pow(gp.mu, (double)(1 + a.iter - gp.iter_mu));   // kappa, gusto.jl:264, 1419-1423
    flags |= (acc ? 1 : 0) | (trust_viol ? 4 : 0) | (feasible ? 8 : 0);
    h[GH_FLAGS] = (double)flags; h[GH_ETANEXT] = eta_n; h[GH_LAMNEXT] = lam_n;
    a.accept[b] = acc ? 1 : 0;
    a.scal[b] = eta_n; a.scal[a.BS + b] = lam_n;
    if (acc) a.J_ref[b] = J_aug;
    if (a.iter >= gp.iter_max) { a.active[b] = 0; return; }
    atomicAdd(a.n_active, 1);
}

__global__ void fill_nan_kernel(double* dst, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) dst[b] = __longlong_as_double(0x7ff8000000000000LL);
}

}  // namespace scp

extern "C" int scp_gusto_init_host(scp_sub_handle s, scp_sub_handle proj, int B, const scp_gusto_params* pars, const double* xd,
                                   const double* ud, const double* p, const double* pp)
{
    if (!s || !pars || B < 1 || !xd || !ud) return SCP_ERR_BAD_ARGUMENT;
    scp_problem* h = s->h;
    if (B > h->cap) { s->err = "batch size exceeds batch_capacity"; return SCP_ERR_BATCH_TOO_LARGE; }
    if ((h->npt > 0 && !p) || (h->info.npp > 0 && !pp)) { s->err = "missing input"; return SCP_ERR_BAD_ARGUMENT; }
    if (pars->iter_max < 1 || pars->nst < 0 || s->nscal != 2 || s->nfun != h->N * (1 + pars->nst)) {
        s->err = "not a GuSTO template (nscal = 2: eta, lambda; fun = v_tr[N], v_st[nst, N])";
        return SCP_ERR_BAD_ARGUMENT;
    }
    if (proj && proj->h != h) { s->err = "projection template belongs to another problem handle"; return SCP_ERR_BAD_ARGUMENT; }
    {   // the solution costs (gusto_post_kernel) penalise exactly: the cone indicators of the model's convex state set X + its s rows
        int nq = 0, ok = 0;
        (void)with_model(h->model_id, [&](auto m) -> int {
            using M = decltype(m);
            nq = scp::count_x_indicators<M>(M::make_params(h->par.data()), h->N);
            ok = M::s_input_free ? 1 : 0;
            return (int)SCP_OK;
        });
        if (!ok) { s->err = "GuSTO: the model's s depends on the input (gusto.jl:757-792 needs s(t, k, x, p))"; return SCP_ERR_UNSUPPORTED; }
        if (pars->nst != nq + h->info.ns) {
            s->err = "GuSTO template: nst must be (cone indicators of X) + ns = " + std::to_string(nq + h->info.ns);
            return SCP_ERR_UNSUPPORTED;
        }
    }
    SUB_TRY(hipSetDevice(h->device));
    int rc;
    if ((rc = sub_loop_state(s, pars->iter_max)) != SCP_OK) return rc;
    if (!(pars->q_exit >= 1.0) || !(pars->q_tr >= 1.0)) { s->err = "q_exit and q_tr must be >= 1 (or Inf)"; return SCP_ERR_BAD_ARGUMENT; }
    if ((pars->pen != 0 && pars->pen != 1) || (pars->pen == 1 && !(pars->hom > 0.0))) { s->err = "pen must be 0 (:quad) or 1 (:softplus, hom > 0)"; return SCP_ERR_BAD_ARGUMENT; }
    if (pars->pen == 1 && s->eng.sched.nexp == 0) { s->err = "pen = :softplus needs a template with exponential cones"; return SCP_ERR_BAD_ARGUMENT; }
    s->gp = *pars; s->B = B; s->iter = 0; s->iter_max = pars->iter_max; s->gusto_ready = true; s->scvx_ready = false; s->ptr_ready = false;
    s->q_exit = pars->q_exit; s->q_tr = pars->q_tr;
    TRY(upload_traj(h, B, xd, ud, p, h->ref_xd, h->ref_ud, h->ref_p));
    if (h->info.npp > 0) {
        SUB_TRY(hipMemcpyAsync(s->d_pp, pp, sizeof(double) * h->info.npp * B, hipMemcpyHostToDevice, h->stream));
        if (proj) SUB_TRY(hipMemcpyAsync(proj->d_pp, pp, sizeof(double) * h->info.npp * B, hipMemcpyHostToDevice, h->stream));
    }
    SUB_TRY(hipMemsetAsync(s->hist, 0, sizeof(double) * (size_t)pars->iter_max * B * SCP_SCVX_HIST_WIDTH, h->stream));
    SUB_TRY(hipMemsetAsync(s->status, 0, sizeof(int) * (size_t)B, h->stream));
    SUB_TRY(hipMemsetAsync(s->iters_done, 0, sizeof(int) * (size_t)B, h->stream));
    std::vector<int> ones(B, 1);
    SUB_TRY(hipMemcpyAsync(s->active, ones.data(), sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    SUB_TRY(hipStreamSynchronize(h->stream));
    if (proj) {   // generate_initial_guess, gusto.jl:516-521 -> correct_convex!, scp.jl:275-361
        TRY(discretize_dev(h, B, h->ref_xd, h->ref_ud, h->ref_p, h->ref_dyn, h->d_feas, nullptr));
        if ((rc = sub_fill_sources(proj, B, nullptr)) != SCP_OK) { s->err = proj->err; return rc; }
        if ((rc = sub_solve_dev(proj, B, sub_opts(&pars->solver), nullptr)) != SCP_OK) { s->err = proj->err; return rc; }
        int* acc = s->active + 3 * (size_t)h->cap + 1;
        hipLaunchKernelGGL(scp::proj_status_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, proj->eng.status, s->active,
                           s->status, acc, B);
        if ((rc = sub_masked_copy_all(s, B, acc, true)) != SCP_OK) return rc;
    }
    TRY(discretize_dev(h, B, h->ref_xd, h->ref_ud, h->ref_p, h->ref_dyn, h->d_feas, nullptr));
    double* scal = s->src + s->lay.off[scp::SEG_SCAL] * s->eng.BS;
    const dim3 g((B + 255) / 256), t(256);
    hipLaunchKernelGGL(scp::fill_nan_kernel, g, t, 0, h->stream, s->J_ref, B);           // ref.J_aug = NaN before the first solve
    hipLaunchKernelGGL(scp::fill_strided_kernel, g, t, 0, h->stream, scal, pars->eta_init, B);
    hipLaunchKernelGGL(scp::fill_strided_kernel, g, t, 0, h->stream, scal + s->eng.BS, pars->lam_init, B);
    SUB_TRY(hipGetLastError());
    SUB_TRY(hipStreamSynchronize(h->stream));
    stamps_collect(h);
    return SCP_OK;
}

extern "C" int scp_gusto_iterate(scp_sub_handle s, int* n_active)
{
    if (!s || !s->gusto_ready) return SCP_ERR_BAD_ARGUMENT;
    scp_problem* h = s->h;
    SUB_TRY(hipSetDevice(h->device));
    if (s->iter >= s->gp.iter_max) { if (n_active) *n_active = 0; return SCP_OK; }
    const int B = s->B;
    s->iter += 1;
    int rc;
    int* acc = s->active + 3 * (size_t)h->cap + 1;
    SUB_TRY(hipMemsetAsync(s->n_active, 0, sizeof(int), h->stream));
    if ((rc = sub_fill_sources(s, B, s->active)) != SCP_OK) return rc;
    if ((rc = sub_solve_dev(s, B, sub_opts(&s->gp.solver), s->active)) != SCP_OK) return rc;
    if ((rc = sub_post(s, B, h->sol_xd, h->sol_ud, h->sol_p, h->sol_dyn.defect, s->active)) != SCP_OK) return rc;
    scp::GustoPostArgs pa;
    pa.B = B; pa.N = h->N; pa.xd = h->sol_xd; pa.ud = h->sol_ud; pa.p = h->sol_p; pa.rxd = h->ref_xd; pa.rud = h->ref_ud;
    pa.rp = h->ref_p; pa.post2 = s->post2; pa.active = s->active; pa.pen = s->gp.pen; pa.hom = s->gp.hom;
    rc = with_model(h->model_id, [&](auto m) -> int {
        using M = decltype(m);
        typename M::Params P = M::make_params(h->par.data());
        hipLaunchKernelGGL(scp::gusto_post_kernel<M>, dim3(B), dim3(64), 0, h->stream, pa, P);
        return (int)SCP_OK;
    });
    if (rc != SCP_OK) return rc;
    scp::GustoUpdateArgs a;
    a.B = B; a.iter = s->iter; a.N = h->N; a.nst = s->gp.nst; a.BS = s->eng.BS; a.gp = s->gp; a.post = s->post; a.post2 = s->post2;
    a.fun = s->funv; a.feas = h->d_feas_new; a.ipm_status = s->eng.status; a.ipm_iters = s->eng.iters; a.J_ref = s->J_ref;
    a.J_last = s->J_ref + h->cap; a.scal = s->src + s->lay.off[scp::SEG_SCAL] * s->eng.BS; a.active = s->active; a.accept = acc;
    a.scp_status = s->status; a.iters_done = s->iters_done; a.hist = s->hist; a.n_active = s->n_active;
    hipLaunchKernelGGL(scp::gusto_update_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, a);
    SUB_TRY(hipGetLastError());
    if ((rc = sub_masked_copy_all(s, B, acc, true)) != SCP_OK) return rc;    // ref = sol for the accepted steps
    int na = 0;
    SUB_TRY(hipMemcpyAsync(&na, s->n_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    SUB_TRY(hipStreamSynchronize(h->stream));
    stamps_collect(h);
    if (n_active) *n_active = na;
    return SCP_OK;
}

extern "C" int scp_gusto_get_host(scp_sub_handle s, double* xd, double* ud, double* p, int32_t* status, int32_t* iterations,
                                  double* cost, uint8_t* feas, double* defect, double* hist)
{
    if (!s || !s->gusto_ready) return SCP_ERR_BAD_ARGUMENT;
    return scp_scvx_get_host(s, xd, ud, p, status, iterations, cost, feas, defect, hist);
}

// -------------------------------------------------------------------------------------------------------------------
// PTR outer loop on the device over ANY PTR template (src/solvers/ptr.jl:448-532): q_tr in {1, 2, 4, Inf}, models without the
// stage-structured fast path (Starship, free-flyer)
// -------------------------------------------------------------------------------------------------------------------
namespace scp {

struct PtrgUpdateArgs {
    int B, iter;
    long BS;
    scp_ptr_generic_params pp;
    const double* post;      // [B][4]: -, -, deviation (q_exit norm)
    const double* fun;       // interleaved: fun[0] = trapz(P) + sum(Pf), fun[1] = trapz(eta_x) + trapz(eta_u) + eta_p
    const double* info;      // interleaved [8][BS] of the conic solve: pcost, dcost, gap, pres, dres, ...
    const int* feas;
    const int* ipm_status;
    const int* ipm_iters;
    double* J_ref;           // [B] J_aug of the reference (NaN for the initial guess, ptr.jl:350)
    double* cost;            // [B][4]: J, J_tr, J_vc, J_aug of the last subproblem
    int* active;
    int* accept;
    int* scp_status;
    int* iters_done;
    double* hist;            // [iter_max][B][SCP_HIST_WIDTH], the columns of scp_ptr_get_host
    int* n_active;
};

// SubproblemSolution cost split (ptr.jl:753-895) + check_stopping_criterion! (:908-932) + ref = spbm.sol (:509)
__global__ void ptrg_update_kernel(PtrgUpdateArgs a)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    a.accept[b] = 0;
    if (!a.active[b]) return;
    double* h = a.hist + ((long)(a.iter - 1) * a.B + b) * SCP_HIST_WIDTH;
    const scp_ptr_generic_params& pp = a.pp;
    const double J_aug = a.info[b] + pp.cost_const;
    const double J_vc = pp.wvc * a.fun[b], J_tr = pp.wtr * a.fun[a.BS + b], J = J_aug - J_vc - J_tr;
    const double dev = a.post[(long)b * 4 + 2];
    const double J_ref = a.J_ref[b];
    const double improv = (J_ref - J_aug) / fabs(J_ref);
    const bool unsafe = a.ipm_status[b] > 1;                     // unsafe_solution, scp.jl:965-980
    const bool feas = a.feas[b] != 0;
    const bool stop = a.iter > 1 && (feas && (fabs(improv) <= pp.eps_rel || dev <= pp.eps_abs));
    h[0] = J; h[1] = J_tr; h[2] = J_vc; h[3] = J_aug; h[4] = dev; h[5] = improv; h[6] = feas ? 1.0 : 0.0;
    h[7] = (double)a.ipm_status[b]; h[8] = (double)a.ipm_iters[b]; h[9] = 1.0;
    h[10] = a.info[2 * a.BS + b]; h[11] = a.info[3 * a.BS + b]; h[12] = a.info[4 * a.BS + b];
    double* c = a.cost + (long)b * 4;
    c[0] = J; c[1] = J_tr; c[2] = J_vc; c[3] = J_aug;
    a.iters_done[b] = a.iter;
    if (unsafe) { a.scp_status[b] = 1; a.active[b] = 0; return; }    // emergency exit before ref = sol (ptr.jl:488-491)
    if (stop) { a.active[b] = 0; return; }                            // `break` BEFORE ref = spbm.sol (ptr.jl:500-509)
    a.accept[b] = 1;                                                  // ref = spbm.sol
    a.J_ref[b] = J_aug;
    if (a.iter >= pp.iter_max) { a.active[b] = 0; return; }
    atomicAdd(a.n_active, 1);
}

}  // namespace scp

extern "C" int scp_ptr_generic_init_host(scp_sub_handle s, int B, const scp_ptr_generic_params* pars, const double* xd,
                                         const double* ud, const double* p, const double* pp)
{
    if (!s || !pars || B < 1 || !xd || !ud) return SCP_ERR_BAD_ARGUMENT;
    scp_problem* h = s->h;
    if (B > h->cap) { s->err = "batch size exceeds batch_capacity"; return SCP_ERR_BATCH_TOO_LARGE; }
    if ((h->npt > 0 && !p) || (h->info.npp > 0 && !pp)) { s->err = "missing input"; return SCP_ERR_BAD_ARGUMENT; }
    if (pars->iter_max < 1 || s->nscal > 1 || s->nfun < 2) { s->err = "not a PTR template (nscal <= 1 and unused, fun[0] = virtual-control penalty, fun[1] = trust-region penalty)"; return SCP_ERR_BAD_ARGUMENT; }
    if (!(pars->q_exit >= 1.0)) { s->err = "q_exit must be >= 1 (or Inf)"; return SCP_ERR_BAD_ARGUMENT; }
    SUB_TRY(hipSetDevice(h->device));
    int rc;
    if ((rc = sub_loop_state(s, pars->iter_max)) != SCP_OK) return rc;
    s->pp_ = *pars; s->B = B; s->iter = 0; s->iter_max = pars->iter_max; s->ptr_ready = true; s->scvx_ready = false; s->gusto_ready = false;
    s->q_exit = pars->q_exit; s->q_tr = pars->q_exit;
    TRY(upload_traj(h, B, xd, ud, p, h->ref_xd, h->ref_ud, h->ref_p));
    if (h->info.npp > 0) SUB_TRY(hipMemcpyAsync(s->d_pp, pp, sizeof(double) * h->info.npp * B, hipMemcpyHostToDevice, h->stream));
    SUB_TRY(hipMemsetAsync(s->hist, 0, sizeof(double) * (size_t)pars->iter_max * B * SCP_HIST_WIDTH, h->stream));
    SUB_TRY(hipMemsetAsync(s->status, 0, sizeof(int) * (size_t)B, h->stream));
    SUB_TRY(hipMemsetAsync(s->iters_done, 0, sizeof(int) * (size_t)B, h->stream));
    SUB_TRY(hipMemsetAsync(s->post2, 0, sizeof(double) * 4 * (size_t)B, h->stream));
    std::vector<int> ones(B, 1);
    SUB_TRY(hipMemcpyAsync(s->active, ones.data(), sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    SUB_TRY(hipStreamSynchronize(h->stream));
    // generate_initial_guess: discretize!(guess) (ptr.jl:548-555); J_aug of the guess is NaN (ptr.jl:350)
    TRY(discretize_dev(h, B, h->ref_xd, h->ref_ud, h->ref_p, h->ref_dyn, h->d_feas, nullptr));
    hipLaunchKernelGGL(scp::fill_nan_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, s->J_ref, B);
    SUB_TRY(hipGetLastError());
    SUB_TRY(hipStreamSynchronize(h->stream));
    stamps_collect(h);
    return SCP_OK;
}

extern "C" int scp_ptr_generic_iterate(scp_sub_handle s, int* n_active)
{
    if (!s || !s->ptr_ready) return SCP_ERR_BAD_ARGUMENT;
    scp_problem* h = s->h;
    SUB_TRY(hipSetDevice(h->device));
    if (s->iter >= s->pp_.iter_max) { if (n_active) *n_active = 0; return SCP_OK; }
    const int B = s->B;
    s->iter += 1;
    int rc;
    int* acc = s->active + 3 * (size_t)h->cap + 1;
    SUB_TRY(hipMemsetAsync(s->n_active, 0, sizeof(int), h->stream));
    if ((rc = sub_fill_sources(s, B, s->active)) != SCP_OK) return rc;
    if ((rc = sub_solve_dev(s, B, sub_opts(&s->pp_.solver), s->active)) != SCP_OK) return rc;
    if ((rc = sub_post(s, B, h->sol_xd, h->sol_ud, h->sol_p, h->sol_dyn.defect, s->active)) != SCP_OK) return rc;
    scp::PtrgUpdateArgs a;
    a.B = B; a.iter = s->iter; a.BS = s->eng.BS; a.pp = s->pp_; a.post = s->post; a.fun = s->funv; a.info = s->eng.info;
    a.feas = h->d_feas_new; a.ipm_status = s->eng.status; a.ipm_iters = s->eng.iters; a.J_ref = s->J_ref; a.cost = s->post2;
    a.active = s->active; a.accept = acc; a.scp_status = s->status; a.iters_done = s->iters_done; a.hist = s->hist;
    a.n_active = s->n_active;
    hipLaunchKernelGGL(scp::ptrg_update_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, a);
    SUB_TRY(hipGetLastError());
    if ((rc = sub_masked_copy_all(s, B, acc, true)) != SCP_OK) return rc;    // ref = spbm.sol (ptr.jl:509)
    int na = 0;
    SUB_TRY(hipMemcpyAsync(&na, s->n_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    SUB_TRY(hipStreamSynchronize(h->stream));
    stamps_collect(h);
    if (n_active) *n_active = na;
    return SCP_OK;
}

extern "C" int scp_ptr_generic_get_host(scp_sub_handle s, double* xd, double* ud, double* p, int32_t* status, int32_t* iterations,
                                        double* cost, uint8_t* feas, double* defect, double* hist)
{
    if (!s || !s->ptr_ready) return SCP_ERR_BAD_ARGUMENT;
    scp_problem* h = s->h;
    SUB_TRY(hipSetDevice(h->device));
    int rc = scp_scvx_get_host(s, xd, ud, p, status, iterations, nullptr, feas, defect, hist);
    if (rc != SCP_OK) return rc;
    if (cost) {
        SUB_TRY(hipMemcpyAsync(cost, s->post2, sizeof(double) * 4 * (size_t)s->B, hipMemcpyDeviceToHost, h->stream));
        SUB_TRY(hipStreamSynchronize(h->stream));
    }
    return SCP_OK;
}
