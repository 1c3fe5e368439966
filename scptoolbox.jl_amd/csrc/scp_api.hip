// C-ABI implementation (include/scp_mi355x.h) of the MI355X-native SCP inner loop.
// gfx950 only: no CUDA shims, no dual paths.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "../../include/scp_mi355x.h"
#include "discretize_kernel.hpp"
#include "ipm_kernel.hpp"
#include "ipm2_kernel.hpp"
#include "models/double_integrator.hpp"
#include "models/quadrotor.hpp"
#include "models/rocket_landing.hpp"
#include "models/starship.hpp"
#include "models/freeflyer.hpp"
#include "ptr_kernels.hpp"
#include "sharded_loop.hpp"
#include "starship_guess.hpp"
#include "stage_problem.hpp"

using namespace scp;

struct DynBuf {  // one DLTV + defect on the device
    double *A = nullptr, *Bm = nullptr, *Bp = nullptr, *F = nullptr, *r = nullptr, *E = nullptr, *defect = nullptr;
};

struct StarshipGuessState;
static void starship_guess_free(StarshipGuessState* g);

struct scp_problem {
    int model_id = -1;
    scp_model_info info{};
    int N = 0, Nsub = 0, method = 0, cap = 0, device = 0;
    int npt = 0;   // length of the parameter vector: info.np global + info.np_node per node (model_common.hpp)
    double feas_tol = 0;
    std::vector<double> par;
    std::vector<double> Sx, cx, Su, cu, Sp, cp;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // per-kernel timing: events recorded around every launch, accumulated at the next stream sync
    struct Stamp { hipEvent_t a, b; int kind; };
    std::vector<Stamp> stamps_free, stamps_pending;
    double t_kernel[4] = {0, 0, 0, 0};
    long n_kernel[4] = {0, 0, 0, 0};
    std::vector<void*> allocs;
    double *guess_xd = nullptr, *guess_ud = nullptr, *guess_p = nullptr;
    double *q_pp = nullptr, *q_xd = nullptr, *q_ud = nullptr, *q_p = nullptr;   // scratch of scp_guess_batch_host (a pure query)
    struct StarshipGuessState* sg = nullptr;   // device-side reference guess of the Starship model (starship_guess.hpp), lazily built
    int guess_failures = 0;                   // instances of the last scp_guess_batch_host call that fell back to the straight line
    long long* prof = nullptr;
    // trajectories
    double *ref_xd = nullptr, *ref_ud = nullptr, *ref_p = nullptr;
    double *sol_xd = nullptr, *sol_ud = nullptr, *sol_p = nullptr;
    DynBuf ref_dyn, sol_dyn;
    double* d_pp = nullptr;
    double *d_iSx = nullptr, *d_Sx = nullptr, *d_cx = nullptr, *d_Su = nullptr, *d_cu = nullptr, *d_Sp = nullptr,
           *d_cp = nullptr;
    int *d_feas_new = nullptr, *d_feas = nullptr;
    int* d_mvar = nullptr;   // [2 cap] per-problem choice of the discretize! form (disc_split_kernel)
    // subproblem
    double *slab = nullptr, *work = nullptr, *z_out = nullptr, *p_out = nullptr, *ipm_info = nullptr, *cost = nullptr,
           *dev = nullptr, *eta = nullptr, *Jaug_ref = nullptr, *hist = nullptr;
    double *vd = nullptr, *vs = nullptr, *vic = nullptr, *vtc = nullptr, *Ppen = nullptr, *Pf = nullptr;   // ptr.jl:399-432
    int *ipm_status = nullptr, *ipm_iters = nullptr, *active = nullptr, *scp_status = nullptr, *iters_done = nullptr,
        *n_active = nullptr, *cold_iters = nullptr, *snap = nullptr;
    long slab_stride = 0, work_stride = 0;
    bool ptr_ready = false;   // subproblem buffers allocated
    bool run_ready = false;   // a PTR run was initialised by scp_ptr_init_host / scp_ptr_init_guess_host (guesses resident)
    bool sub_ready = false;   // a subproblem has been solved (virtual controls available)
    int num_cus = 256;      // multiProcessorCount of the device (set at create)
    int wpe_override = std::getenv("SCP_IPM_WPE") ? std::atoi(std::getenv("SCP_IPM_WPE")) : 0;   // tuning aid
    // debugging / parity aid: force the reference formulation of discretize! (K1) for const-Jacobian models too
    bool disc_reference_form = std::getenv("SCP_DISC_REFERENCE_FORM") != nullptr;
    int disc_bits = 64;     // arithmetic of discretize! (scp_set_discretize_precision): 64 = reference, 32 = tolerance check
    // PTR run state
    scp_ptr_params pars{};
    int B = 0, iter = 0, hist_cap = 0;
    int na_cap = 0;
    int* na_ring = nullptr;              // pinned host copy of n_active after every enqueued iteration (scp_ptr_poll_iteration)
    int* na_dev = nullptr;               // the same ring ON THE DEVICE: what the multi-GPU all-reduce sums (scp_ptr_run_sharded)
    std::vector<hipEvent_t> na_ev;       // na_ev[k]: recorded behind the copy of iteration k
    std::string err;
};

#define HIP_TRY(h, call)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            if (h) (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);             \
            return SCP_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

template <class M>
static void fill_info(scp_model_info* i)
{
    std::memset(i, 0, sizeof(*i));
    i->nx = M::nx; i->nu = M::nu; i->np = M::np; i->npF = M::npF;
    for (int j = 0; j < M::npF && j < 8; j++) i->Fcols[j] = M::Fcol(j);
    i->ns = M::ns; i->nic = M::nic; i->ntc = M::ntc; i->npar = M::npar; i->npp = M::npp;
    i->nl = M::nl; i->nsoc = M::nsoc; i->ng = M::ng;
    i->structured = M::structured ? 1 : 0;
    i->has_subproblem = M::has_subproblem ? 1 : 0;
    i->np_node = M::np_node;
    i->global_rows_in_X = M::global_rows_in_X ? 1 : 0;
    i->linf_groups = M::linf_groups; i->linf_rows = M::linf_rows;
    i->s_input_free = M::s_input_free ? 1 : 0;
}

// dispatch a generic lambda on the model type
template <class Fn>
static int with_model(int model_id, Fn&& fn)
{
    switch (model_id) {
        case SCP_MODEL_DOUBLE_INTEGRATOR: return fn(DoubleIntegrator{});
        case SCP_MODEL_QUADROTOR: return fn(Quadrotor{});
        case SCP_MODEL_ROCKET_LANDING: return fn(RocketLanding{});
        case SCP_MODEL_STARSHIP: return fn(Starship{});
        case SCP_MODEL_FREEFLYER: return fn(Freeflyer{});
        default: return SCP_ERR_UNKNOWN_MODEL;
    }
}
// models with the stage-structured PTR fast path (stage_problem.hpp + ipm2_*.hpp: one arrow column, <= 16 penalised rows
// per node); the others (M::structured == false) run their subproblems through the generic conic path
template <class Fn>
static int with_structured_model(int model_id, Fn&& fn)
{
    switch (model_id) {
        case SCP_MODEL_DOUBLE_INTEGRATOR: return fn(DoubleIntegrator{});
        case SCP_MODEL_QUADROTOR: return fn(Quadrotor{});
        case SCP_MODEL_ROCKET_LANDING: return fn(RocketLanding{});
        case SCP_MODEL_STARSHIP: return SCP_ERR_UNSUPPORTED;
        case SCP_MODEL_FREEFLYER: return SCP_ERR_UNSUPPORTED;
        default: return SCP_ERR_UNKNOWN_MODEL;
    }
}

extern "C" int scp_model_query(int model_id, scp_model_info* info)
{
    if (!info) return SCP_ERR_BAD_ARGUMENT;
    return with_model(model_id, [&](auto m) { fill_info<decltype(m)>(info); return (int)SCP_OK; });
}

// Host-side evaluation of the compiled model's convex sets and cost (the X / U / cost closures of TrajectoryProblem).
extern "C" int scp_model_rows(int model_id, const double* model_par, int N, int k, double* L, double* Lp, double* l,
                              double* Mm, double* m, double* Lg, double* lg, double* cost)
{
    if (!model_par || N < 2 || k < 1 || k > N) return SCP_ERR_BAD_ARGUMENT;
    return with_model(model_id, [&](auto mt) {
        using M = decltype(mt);
        constexpr int nx = M::nx, nu = M::nu, np = M::np, npa = np > 0 ? np : 1, nz = nx + nu, npc = np_compact<M>(),
                      npca = npc > 0 ? npc : 1;
        const typename M::Params P = M::make_params(model_par);
        const double t = (1.0 - (double)(k - 1) / (double)(N - 1)) * 0.0 + ((double)(k - 1) / (double)(N - 1)) * 1.0;
        if constexpr (M::nl > 0) {
            double Lb[M::nl * nz], Lpb[M::nl * npca], lb[M::nl];
            for (int i = 0; i < M::nl * npca; i++) Lpb[i] = 0.0;
            M::lin_rows(P, t, k, Lb, Lpb, lb);
            if (L) std::memcpy(L, Lb, sizeof(Lb));
            if (Lp) for (int i = 0; i < M::nl; i++) for (int j = 0; j < npc; j++) Lp[i * npc + j] = Lpb[i * npca + j];
            if (l) std::memcpy(l, lb, sizeof(lb));
        }
        if constexpr (M::nsoc > 0) {
            double Mb[M::nsoc * 4 * nz], mb[M::nsoc * 4];
            M::soc_rows(P, t, k, Mb, mb);
            if (Mm) std::memcpy(Mm, Mb, sizeof(Mb));
            if (m) std::memcpy(m, mb, sizeof(mb));
        }
        if constexpr (M::ng > 0) {
            double Lgb[M::ng * npa], lgb[M::ng];
            M::glin_rows(P, Lgb, lgb);
            if (Lg) for (int i = 0; i < M::ng; i++) for (int j = 0; j < np; j++) Lg[i * np + j] = Lgb[i * npa + j];
            if (lg) std::memcpy(lg, lgb, sizeof(lgb));
        }
        if (cost) {
            double Qu[nu], lu[nu], lx[nx], tx[nx], tp[npca], Qp[npca];
            for (int i = 0; i < npca; i++) { tp[i] = 0.0; Qp[i] = 0.0; }
            M::cost_terms(P, Qu, lu, lx, tx, tp, Qp);
            double* c = cost;
            for (int i = 0; i < nu; i++) *c++ = Qu[i];
            for (int i = 0; i < nu; i++) *c++ = lu[i];
            for (int i = 0; i < nx; i++) *c++ = lx[i];
            for (int i = 0; i < nx; i++) *c++ = tx[i];
            for (int i = 0; i < npc; i++) *c++ = tp[i];
            for (int i = 0; i < npc; i++) *c++ = Qp[i];
        }
        return (int)SCP_OK;
    });
}

// number of cone indicators of the convex state set X per node (GuSTO's soft penalties, gusto.jl:883-934): what
// scp_gusto_init_host expects as nst - ns
extern "C" int scp_model_state_indicators(int model_id, const double* model_par, int N, int* nq)
{
    if (!model_par || N < 2 || !nq) return SCP_ERR_BAD_ARGUMENT;
    return with_model(model_id, [&](auto mt) {
        using M = decltype(mt);
        *nq = count_x_indicators<M>(M::make_params(model_par), N);
        return (int)SCP_OK;
    });
}

// Host-side evaluation of the compiled model's closures at one point -- the counterpart of calling traj.f/A/B/F
// (problem.jl:432-450), traj.s/C/D/G (:560-600) and the cone indicators of X from Julia: lets a maintainer (and the CPU
// tests) check a compiled model against the closures it replaces without a GPU.
extern "C" int scp_model_eval_host(int model_id, const double* model_par, int N, int k, const double* x, const double* u,
                                   const double* p, double* f, double* A, double* B, double* F, double* s, double* C, double* D,
                                   double* G, double* q, int* nq)
{
    if (!model_par || N < 2 || k < 1 || k > N || !x || !u) return SCP_ERR_BAD_ARGUMENT;
    return with_model(model_id, [&](auto mt) {
        using M = decltype(mt);
        constexpr int nx = M::nx, nu = M::nu, npF = M::npF, npFa = npF > 0 ? npF : 1, ns = M::ns, nsa = ns > 0 ? ns : 1,
                      npc = np_compact<M>(), npca = npc > 0 ? npc : 1;
        if (np_total<M>(N) > 0 && !p) return (int)SCP_ERR_BAD_ARGUMENT;
        const typename M::Params P = M::make_params(model_par);
        const double t = (1.0 - (double)(k - 1) / (double)(N - 1)) * 0.0 + ((double)(k - 1) / (double)(N - 1)) * 1.0;
        double xs[nx], us[nu];
        for (int i = 0; i < nx; i++) xs[i] = x[i];
        for (int i = 0; i < nu; i++) us[i] = u[i];
        if (f || A || B || F) {
            double fb[nx], Ab[nx * nx], Bb[nx * nu], Fb[nx * npFa];
            M::dyn(P, t, k, xs, us, p, fb, Ab, Bb, Fb);
            if (f) std::memcpy(f, fb, sizeof(fb));
            if (A) std::memcpy(A, Ab, sizeof(Ab));
            if (B) std::memcpy(B, Bb, sizeof(Bb));
            if (F && npF > 0) std::memcpy(F, Fb, sizeof(double) * nx * npF);
        }
        if constexpr (ns > 0) {
            if (s || C || D || G) {
                double sb[nsa], Cb[nsa * nx], Db[nsa * nu], Gb[nsa * npca];
                for (int i = 0; i < nsa * npca; i++) Gb[i] = 0.0;
                M::s_eval(P, t, k, x, u, p, sb, Cb, Db, Gb);
                if (s) std::memcpy(s, sb, sizeof(double) * ns);
                if (C) std::memcpy(C, Cb, sizeof(double) * ns * nx);
                if (D) std::memcpy(D, Db, sizeof(double) * ns * nu);
                if (G) for (int i = 0; i < ns; i++) for (int j = 0; j < npc; j++) G[i * npc + j] = Gb[i * npca + j];
            }
        }
        int n = 0;
        for_each_x_indicator<M>(P, t, k, x, p, N, [&](double v) { if (q) q[n] = v; n++; });
        if (nq) *nq = n;
        return (int)SCP_OK;
    });
}

extern "C" const char* scp_last_error(scp_handle h) { return h ? h->err.c_str() : "null handle"; }

static int stamp_begin(scp_problem* h, int kind)
{
    scp_problem::Stamp st;
    if (!h->stamps_free.empty()) { st = h->stamps_free.back(); h->stamps_free.pop_back(); }
    else { HIP_TRY(h, hipEventCreate(&st.a)); HIP_TRY(h, hipEventCreate(&st.b)); }
    st.kind = kind;
    HIP_TRY(h, hipEventRecord(st.a, h->stream));
    h->stamps_pending.push_back(st);
    return SCP_OK;
}
static int stamp_end(scp_problem* h)
{
    HIP_TRY(h, hipEventRecord(h->stamps_pending.back().b, h->stream));
    return SCP_OK;
}
// call after a stream synchronise
static void stamps_collect(scp_problem* h)
{
    for (auto& st : h->stamps_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, st.a, st.b) == hipSuccess) { h->t_kernel[st.kind] += ms * 1e-3; h->n_kernel[st.kind] += 1; }
        h->stamps_free.push_back(st);
    }
    h->stamps_pending.clear();
}

// without waiting: the stamps at the head of the list whose end event has completed (scp_ptr_poll_iteration)
static void stamps_collect_ready(scp_problem* h)
{
    size_t n = 0;
    while (n < h->stamps_pending.size() && hipEventQuery(h->stamps_pending[n].b) == hipSuccess) {
        auto& st = h->stamps_pending[n];
        float ms = 0;
        if (hipEventElapsedTime(&ms, st.a, st.b) == hipSuccess) { h->t_kernel[st.kind] += ms * 1e-3; h->n_kernel[st.kind] += 1; }
        h->stamps_free.push_back(st);
        n++;
    }
    h->stamps_pending.erase(h->stamps_pending.begin(), h->stamps_pending.begin() + (long)n);
}

template <class T>
static int dalloc(scp_problem* h, T** p, size_t count)
{
    void* v = nullptr;
    HIP_TRY(h, hipMalloc(&v, (count > 0 ? count : 1) * sizeof(T)));
    h->allocs.push_back(v);
    *p = (T*)v;
    return SCP_OK;
}
#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

static int alloc_dyn(scp_problem* h, DynBuf& d)
{
    const size_t nx = h->info.nx, nu = h->info.nu, npF = h->info.npF > 0 ? h->info.npF : 1, M = h->N - 1, B = h->cap;
    TRY(dalloc(h, &d.A, nx * nx * M * B)); TRY(dalloc(h, &d.Bm, nx * nu * M * B)); TRY(dalloc(h, &d.Bp, nx * nu * M * B));
    TRY(dalloc(h, &d.F, nx * npF * M * B)); TRY(dalloc(h, &d.r, nx * M * B)); TRY(dalloc(h, &d.E, nx * nx * M * B));
    TRY(dalloc(h, &d.defect, nx * M * B));
    return SCP_OK;
}

extern "C" int scp_problem_create(const scp_problem_desc* d, scp_handle* out)
{
    if (!d || !out) return SCP_ERR_BAD_ARGUMENT;
    *out = nullptr;
    scp_model_info info;
    int rc = scp_model_query(d->model_id, &info);
    if (rc) return rc;
    if (d->N < 2 || d->Nsub < 2 || d->batch_capacity < 1) return SCP_ERR_BAD_ARGUMENT;
    if (d->disc_method != SCP_FOH && d->disc_method != SCP_IMPULSE) return SCP_ERR_BAD_ARGUMENT;
    if (d->disc_method == SCP_IMPULSE) {   // the model must define its impulse response (f, B evaluated with k < 0)
        const int rci = with_model(d->model_id, [&](auto mt) { return decltype(mt)::has_impulse ? (int)SCP_OK : (int)SCP_ERR_UNSUPPORTED; });
        if (rci != SCP_OK) return rci;
    }
    if (!d->model_par || !d->scale.Sx || !d->scale.cx || !d->scale.Su || !d->scale.cu) return SCP_ERR_BAD_ARGUMENT;
    const int npt = info.np + info.np_node * d->N;
    if (npt > 0 && (!d->scale.Sp || !d->scale.cp)) return SCP_ERR_BAD_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SCP_ERR_NO_DEVICE;
    if (d->device < 0 || d->device >= ndev) return SCP_ERR_NO_DEVICE;
    scp_problem* h = new (std::nothrow) scp_problem();
    if (!h) return SCP_ERR_ALLOC;
    h->model_id = d->model_id; h->info = info; h->N = d->N; h->Nsub = d->Nsub; h->method = d->disc_method;
    h->cap = d->batch_capacity; h->device = d->device; h->feas_tol = d->feas_tol; h->npt = npt;
    h->par.assign(d->model_par, d->model_par + info.npar);
    h->Sx.assign(d->scale.Sx, d->scale.Sx + info.nx); h->cx.assign(d->scale.cx, d->scale.cx + info.nx);
    h->Su.assign(d->scale.Su, d->scale.Su + info.nu); h->cu.assign(d->scale.cu, d->scale.cu + info.nu);
    if (npt > 0) {
        h->Sp.assign(d->scale.Sp, d->scale.Sp + npt); h->cp.assign(d->scale.cp, d->scale.cp + npt);
    }
    *out = h;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && ncu > 0) h->num_cus = ncu;
    }
    HIP_TRY(h, hipEventCreate(&h->ev0));
    HIP_TRY(h, hipEventCreate(&h->ev1));
    const size_t nx = info.nx, nu = info.nu, np = npt > 0 ? npt : 1;
    const size_t B = h->cap, N = h->N;
    TRY(dalloc(h, &h->ref_xd, nx * N * B)); TRY(dalloc(h, &h->ref_ud, nu * N * B)); TRY(dalloc(h, &h->ref_p, np * B));
    TRY(dalloc(h, &h->sol_xd, nx * N * B)); TRY(dalloc(h, &h->sol_ud, nu * N * B)); TRY(dalloc(h, &h->sol_p, np * B));
    TRY(alloc_dyn(h, h->ref_dyn)); TRY(alloc_dyn(h, h->sol_dyn));
    TRY(dalloc(h, &h->d_feas_new, B)); TRY(dalloc(h, &h->d_feas, B));
    TRY(dalloc(h, &h->d_iSx, nx)); TRY(dalloc(h, &h->d_Sx, nx)); TRY(dalloc(h, &h->d_cx, nx));
    TRY(dalloc(h, &h->d_Su, nu)); TRY(dalloc(h, &h->d_cu, nu)); TRY(dalloc(h, &h->d_Sp, np)); TRY(dalloc(h, &h->d_cp, np));
    std::vector<double> iSx(nx);
    for (size_t i = 0; i < nx; i++) iSx[i] = 1.0 / h->Sx[i];  // iSx = inv(Sx), scp.jl:492-493
    const size_t D = sizeof(double);
    HIP_TRY(h, hipMemcpy(h->d_iSx, iSx.data(), nx * D, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_Sx, h->Sx.data(), nx * D, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_cx, h->cx.data(), nx * D, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_Su, h->Su.data(), nu * D, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_cu, h->cu.data(), nu * D, hipMemcpyHostToDevice));
    if (npt > 0) {
        HIP_TRY(h, hipMemcpy(h->d_Sp, h->Sp.data(), (size_t)npt * D, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_cp, h->cp.data(), (size_t)npt * D, hipMemcpyHostToDevice));
    }
    return SCP_OK;
}

extern "C" int scp_problem_destroy(scp_handle h)
{
    if (!h) return SCP_ERR_BAD_ARGUMENT;
    (void)hipSetDevice(h->device);
    if (h->sg) starship_guess_free(h->sg);
    for (void* p : h->allocs) (void)hipFree(p);
    for (auto& st : h->stamps_free) { (void)hipEventDestroy(st.a); (void)hipEventDestroy(st.b); }
    for (auto& st : h->stamps_pending) { (void)hipEventDestroy(st.a); (void)hipEventDestroy(st.b); }
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->na_ev) (void)hipEventDestroy(e);
    if (h->na_ring) (void)hipHostFree(h->na_ring);
    if (h->na_dev) (void)hipFree(h->na_dev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return SCP_OK;
}

extern "C" int scp_set_stream_priority(scp_handle h, int level)
{
    if (!h) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    int least = 0, greatest = 0;      // numerically: greatest priority <= least priority
    HIP_TRY(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
    int pr = -level;                  // level > 0 = higher priority = numerically lower
    pr = std::max(greatest, std::min(least, pr));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    stamps_collect(h);
    hipStream_t ns = nullptr;
    HIP_TRY(h, hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, pr));
    (void)hipStreamDestroy(h->stream);
    h->stream = ns;
    return SCP_OK;
}

extern "C" int scp_sync(scp_handle h)
{
    if (!h) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    stamps_collect(h);
    return SCP_OK;
}

// ------------------------------------------------------------------------------------------
// discretize!
// ------------------------------------------------------------------------------------------

static int discretize_dev(scp_problem* h, int B, const double* xd, const double* ud, const double* p, const DynBuf& d,
                          int* feas, const int* mask)
{
    DiscArgs a;
    a.B = B; a.N = h->N; a.Nsub = h->Nsub;
    a.xd = xd; a.ud = ud; a.p = p; a.iSx = h->d_iSx; a.feas_tol = h->feas_tol;
    a.A = d.A; a.Bm = d.Bm; a.Bp = d.Bp; a.F = d.F; a.r = d.r; a.E = d.E; a.defect = d.defect; a.feas = feas; a.mask = mask;
    HIP_TRY(h, hipMemsetAsync(feas, 0xff, (size_t)B * sizeof(int), h->stream));  // ref.feas = true (:179); non-zero == true
    return with_model(h->model_id, [&](auto m) -> int {
        using M = decltype(m);
        using L = DiscLayout<M>;
        const long groups = (long)a.B * (a.N - 1);
        const int blocks = (int)((groups + L::GROUPS_PER_BLOCK - 1) / L::GROUPS_PER_BLOCK);
        typename M::Params P = M::make_params(h->par.data());
        TRY(stamp_begin(h, 0));
        const double rk4_step = 1.0 / ((double)(a.N - 1) * (double)(a.Nsub - 1));
        if (h->disc_bits == 32) {     // fp32 arithmetic (tolerance-check variant; FOH, models with M::has_fp32)
            if constexpr (M::has_fp32) {
                hipLaunchKernelGGL((discretize_foh_kernel<M, false, float>), dim3(blocks), dim3(256), 0, h->stream, a, P);
            } else {
                return (int)SCP_ERR_UNSUPPORTED;
            }
        } else if (h->method == SCP_IMPULSE) {
            hipLaunchKernelGGL((discretize_foh_kernel<M, true>), dim3(blocks), dim3(256), 0, h->stream, a, P);
        } else if (M::const_jacobian && !h->disc_reference_form && rk4_step <= M::var_form_max_step) {
            // variational form (K1v): thread per (problem, interval, column), blockIdx.y = column
            const unsigned gx = (unsigned)((groups + 255) / 256);
            hipLaunchKernelGGL((discretize_foh_var_kernel<M, false>), dim3(gx, 2 * M::nx + 2 * M::nu), dim3(256), 0, h->stream, a, P);
            hipLaunchKernelGGL((discretize_foh_var_kernel<M, true>), dim3(gx, M::npF + 1), dim3(256), 0, h->stream, a, P);
        } else if (!M::const_jacobian && M::var_form_max_phys_step > 0.0) {
            // state-dependent Jacobians: per problem the variational form (K1x) where it meets the reference formulation to
            // 1e-10 (physical RK4 step below the model's bound), the reference form (K1) elsewhere
            if (!h->d_mvar) { TRY(dalloc(h, &h->d_mvar, 2 * (size_t)h->cap)); }
            int* mvar = h->d_mvar; int* mref = h->d_mvar + h->cap;
            hipLaunchKernelGGL(disc_split_kernel<M>, dim3((a.B + 255) / 256), dim3(256), 0, h->stream, a.B, a.N, a.Nsub, a.p, a.mask, P,
                               h->disc_reference_form ? 1 : 0, mvar, mref);
            DiscArgs av = a; av.mask = mvar;
            const unsigned gx = (unsigned)((groups + 255) / 256);
            hipLaunchKernelGGL((discretize_foh_varx_kernel<M, R_PHI>), dim3(gx, M::nx), dim3(256), 0, h->stream, av, P);
            hipLaunchKernelGGL((discretize_foh_varx_kernel<M, R_BM>), dim3(gx, M::nu), dim3(256), 0, h->stream, av, P);
            hipLaunchKernelGGL((discretize_foh_varx_kernel<M, R_BP>), dim3(gx, M::nu), dim3(256), 0, h->stream, av, P);
            if (M::npF > 0) hipLaunchKernelGGL((discretize_foh_varx_kernel<M, R_F>), dim3(gx, M::npF), dim3(256), 0, h->stream, av, P);
            hipLaunchKernelGGL((discretize_foh_varx_kernel<M, R_R>), dim3(gx, 1), dim3(256), 0, h->stream, av, P);
            hipLaunchKernelGGL((discretize_foh_varx_kernel<M, R_E>), dim3(gx, M::nx), dim3(256), 0, h->stream, av, P);
            DiscArgs ar = a; ar.mask = mref;
            hipLaunchKernelGGL(discretize_foh_kernel<M>, dim3(blocks), dim3(256), 0, h->stream, ar, P);
        } else {
            hipLaunchKernelGGL(discretize_foh_kernel<M>, dim3(blocks), dim3(256), 0, h->stream, a, P);
        }
        TRY(stamp_end(h));
        HIP_TRY(h, hipGetLastError());
        return (int)SCP_OK;
    });
}

extern "C" int scp_set_discretize_precision(scp_handle h, int bits)
{
    if (!h || (bits != 32 && bits != 64)) return SCP_ERR_BAD_ARGUMENT;
    if (bits == 32) {
        const bool ok = with_model(h->model_id, [&](auto m) -> int { return decltype(m)::has_fp32 ? 1 : 0; }) == 1;
        if (!ok || h->method != SCP_FOH) { h->err = "fp32 discretize!: FOH and models with an fp32 evaluation only (starship)"; return SCP_ERR_UNSUPPORTED; }
    }
    h->disc_bits = bits;
    return SCP_OK;
}

extern "C" int scp_discretize_batch_dev(scp_handle h, int B, const double* xd, const double* ud, const double* p,
                                        double* A, double* Bm, double* Bp, double* F, double* r, double* E,
                                        double* defect, int32_t* feas)
{
    if (!h || B < 1 || !xd || !ud || !A || !Bm || !Bp || !F || !r || !E || !defect || !feas) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    DynBuf d;
    d.A = A; d.Bm = Bm; d.Bp = Bp; d.F = F; d.r = r; d.E = E; d.defect = defect;
    return discretize_dev(h, B, xd, ud, p, d, feas, nullptr);
}

static int copy_dyn_out(scp_problem* h, int B, const DynBuf& d, double* A, double* Bm, double* Bp, double* F, double* r,
                        double* E, double* defect)
{
    const size_t nx = h->info.nx, nu = h->info.nu, npF = h->info.npF, M = h->N - 1, D = sizeof(double), b = B;
    if (A) HIP_TRY(h, hipMemcpyAsync(A, d.A, nx * nx * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (Bm) HIP_TRY(h, hipMemcpyAsync(Bm, d.Bm, nx * nu * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (Bp) HIP_TRY(h, hipMemcpyAsync(Bp, d.Bp, nx * nu * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (F && npF > 0) HIP_TRY(h, hipMemcpyAsync(F, d.F, nx * npF * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (r) HIP_TRY(h, hipMemcpyAsync(r, d.r, nx * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (E) HIP_TRY(h, hipMemcpyAsync(E, d.E, nx * nx * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (defect) HIP_TRY(h, hipMemcpyAsync(defect, d.defect, nx * M * b * D, hipMemcpyDeviceToHost, h->stream));
    return SCP_OK;
}

static int upload_traj(scp_problem* h, int B, const double* xd, const double* ud, const double* p, double* dxd,
                       double* dud, double* dp)
{
    const size_t nx = h->info.nx, nu = h->info.nu, np = h->npt, N = h->N, D = sizeof(double), b = B;
    HIP_TRY(h, hipMemcpyAsync(dxd, xd, nx * N * b * D, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(dud, ud, nu * N * b * D, hipMemcpyHostToDevice, h->stream));
    if (np > 0) HIP_TRY(h, hipMemcpyAsync(dp, p, np * b * D, hipMemcpyHostToDevice, h->stream));
    return SCP_OK;
}

static int feas_out(scp_problem* h, int B, const int* dfeas, uint8_t* feas)
{
    std::vector<int> hf(B);
    HIP_TRY(h, hipMemcpyAsync(hf.data(), dfeas, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    stamps_collect(h);
    if (feas)
        for (int i = 0; i < B; i++) feas[i] = hf[i] != 0;
    return SCP_OK;
}

extern "C" int scp_discretize_batch_host(scp_handle h, int B, const double* xd, const double* ud, const double* p,
                                         double* A, double* Bm, double* Bp, double* F, double* r, double* E,
                                         double* defect, uint8_t* feas, double* seconds)
{
    if (!h || B < 1 || !xd || !ud) return SCP_ERR_BAD_ARGUMENT;
    if (B > h->cap) return SCP_ERR_BATCH_TOO_LARGE;
    if (h->npt > 0 && !p) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(upload_traj(h, B, xd, ud, p, h->sol_xd, h->sol_ud, h->sol_p));
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    TRY(discretize_dev(h, B, h->sol_xd, h->sol_ud, h->sol_p, h->sol_dyn, h->d_feas_new, nullptr));
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    TRY(copy_dyn_out(h, B, h->sol_dyn, A, Bm, Bp, F, r, E, defect));
    TRY(feas_out(h, B, h->d_feas_new, feas));
    if (seconds) {
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        *seconds = ms * 1e-3;
    }
    return SCP_OK;
}

extern "C" int scp_propagate_batch_host(scp_handle h, int B, const double* xd, const double* ud, const double* p, int res,
                                        double* xc)
{
    if (!h || B < 1 || !xd || !ud || !xc || res < 2) return SCP_ERR_BAD_ARGUMENT;
    if (B > h->cap) return SCP_ERR_BATCH_TOO_LARGE;
    if (h->npt > 0 && !p) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(upload_traj(h, B, xd, ud, p, h->sol_xd, h->sol_ud, h->sol_p));
    double* d_xc = nullptr;   // result buffer of this call only (post-processing path, not resident)
    const bool imp = h->method == SCP_IMPULSE;
    const int sub = (res + (h->N - 1) - 1) / (h->N - 1);                  // IMPULSE: subres = ceil(res / (N - 1))  (:544)
    const size_t nsamp = imp ? 1 + (size_t)(h->N - 1) * sub : (size_t)res;
    const size_t n = (size_t)h->info.nx * nsamp * (size_t)B;
    HIP_TRY(h, hipMalloc(&d_xc, n * sizeof(double)));
    PropArgs a;
    a.B = B; a.N = h->N; a.res = res; a.xd = h->sol_xd; a.ud = h->sol_ud; a.p = h->sol_p; a.xc = d_xc;
    PropImpArgs ai;
    ai.B = B; ai.N = h->N; ai.sub = sub; ai.xd = h->sol_xd; ai.ud = h->sol_ud; ai.p = h->sol_p; ai.xc = d_xc;
    int rc = with_model(h->model_id, [&](auto m) -> int {
        using M = decltype(m);
        typename M::Params P = M::make_params(h->par.data());
        if (imp) {
            const long tot = (long)B * (h->N - 1);
            hipLaunchKernelGGL(propagate_impulse_kernel<M>, dim3((unsigned)((tot + 63) / 64)), dim3(64), 0, h->stream, ai, P);
        } else {
            hipLaunchKernelGGL(propagate_foh_kernel<M>, dim3((B + 63) / 64), dim3(64), 0, h->stream, a, P);
        }
        return (int)SCP_OK;
    });
    hipError_t e = hipGetLastError();
    if (rc == SCP_OK && e == hipSuccess) e = hipMemcpyAsync(xc, d_xc, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (rc == SCP_OK && e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_xc);
    if (rc != SCP_OK) return rc;
    HIP_TRY(h, e);
    return SCP_OK;
}

// ------------------------------------------------------------------------------------------
// PTR
// ------------------------------------------------------------------------------------------

static int ensure_ptr_buffers(scp_problem* h, int hist_iters)
{
    const size_t B = h->cap;
    if (!h->ptr_ready) {
        int rc = with_structured_model(h->model_id, [&](auto m) -> int {
            using M = decltype(m);
            h->slab_stride = SP<M>::offsets(h->N).total;
            h->work_stride = Ipm2Work<M>::offsets(h->N).total;
            return (int)SCP_OK;
        });
        if (rc) return rc;
        const size_t nz = h->info.nx + h->info.nu, npa = h->npt > 0 ? h->npt : 1, N = h->N;
        if (!h->d_pp) TRY(dalloc(h, &h->d_pp, (size_t)(h->info.npp > 0 ? h->info.npp : 1) * B));
        TRY(dalloc(h, &h->prof, 8 * B));
        TRY(dalloc(h, &h->guess_xd, (size_t)h->info.nx * h->N * B)); TRY(dalloc(h, &h->guess_ud, (size_t)h->info.nu * h->N * B));
        TRY(dalloc(h, &h->guess_p, (size_t)(h->npt > 0 ? h->npt : 1) * B));
        TRY(dalloc(h, &h->slab, (size_t)h->slab_stride * B));
        TRY(dalloc(h, &h->work, (size_t)h->work_stride * B));
        TRY(dalloc(h, &h->z_out, nz * N * B)); TRY(dalloc(h, &h->p_out, npa * B)); TRY(dalloc(h, &h->ipm_info, 8 * B));
        TRY(dalloc(h, &h->cost, 4 * B)); TRY(dalloc(h, &h->dev, B)); TRY(dalloc(h, &h->eta, (2 * N + 1) * B));
        TRY(dalloc(h, &h->Jaug_ref, B));
        TRY(dalloc(h, &h->vd, (size_t)h->info.nx * (N - 1) * B)); TRY(dalloc(h, &h->vs, (size_t)(h->info.ns > 0 ? h->info.ns : 1) * N * B));
        TRY(dalloc(h, &h->vic, (size_t)(h->info.nic > 0 ? h->info.nic : 1) * B)); TRY(dalloc(h, &h->vtc, (size_t)(h->info.ntc > 0 ? h->info.ntc : 1) * B));
        TRY(dalloc(h, &h->Ppen, N * B)); TRY(dalloc(h, &h->Pf, 2 * B));
        TRY(dalloc(h, &h->ipm_status, B)); TRY(dalloc(h, &h->ipm_iters, B)); TRY(dalloc(h, &h->active, B));
        TRY(dalloc(h, &h->scp_status, B)); TRY(dalloc(h, &h->iters_done, B)); TRY(dalloc(h, &h->n_active, 1));
        TRY(dalloc(h, &h->cold_iters, B));
        TRY(dalloc(h, &h->snap, B));
        h->ptr_ready = true;
    }
    if (hist_iters > h->hist_cap) {
        TRY(dalloc(h, &h->hist, (size_t)hist_iters * B * H_N));  // (older, smaller buffer is freed at destroy)
        h->hist_cap = hist_iters;
    }
    return SCP_OK;
}

static int check_pars(const scp_ptr_params* p)
{
    if (!p || p->iter_max < 1 || !(p->wvc > 0) || !(p->wtr > 0)) return SCP_ERR_BAD_ARGUMENT;
    if (!std::isinf(p->q_tr) || !std::isinf(p->q_exit)) return SCP_ERR_UNSUPPORTED;  // reference tests use Inf only
    if (p->ipm_max_iter < 1) return SCP_ERR_BAD_ARGUMENT;
    if (p->ipm_warm != 0 && !(p->ipm_warm_mu > 0.0)) return SCP_ERR_BAD_ARGUMENT;
    return SCP_OK;
}

// formulate (K2) + solve (K3) + extract (K4a) about (ref trajectory, ref_dyn); results in sol_*
static int subproblem_dev(scp_problem* h, int B)
{
    return with_structured_model(h->model_id, [&](auto m) -> int {
        using M = decltype(m);
        typename M::Params P = M::make_params(h->par.data());
        AsmArgs aa;
        aa.B = B; aa.N = h->N; aa.wvc = h->pars.wvc; aa.wtr = h->pars.wtr;
        aa.xd = h->ref_xd; aa.ud = h->ref_ud; aa.p = h->ref_p; aa.pp = h->d_pp;
        aa.A = h->ref_dyn.A; aa.Bm = h->ref_dyn.Bm; aa.Bp = h->ref_dyn.Bp; aa.F = h->ref_dyn.F; aa.r = h->ref_dyn.r;
        aa.Sx = h->d_Sx; aa.cx = h->d_cx; aa.Su = h->d_Su; aa.cu = h->d_cu; aa.Sp = h->d_Sp; aa.cp = h->d_cp;
        aa.slab = h->slab; aa.slab_stride = h->slab_stride; aa.active = h->active;
        const long nthreads = (long)B * (h->N + 1);
        TRY(stamp_begin(h, 1));
        hipLaunchKernelGGL(ptr_assemble_kernel<M>, dim3((unsigned)((nthreads + 63) / 64)), dim3(64), 0, h->stream, aa, P);
        TRY(stamp_end(h));
        HIP_TRY(h, hipGetLastError());
        IpmArgs ia;
        ia.B = B; ia.N = h->N; ia.max_iter = h->pars.ipm_max_iter; ia.nref = h->pars.ipm_nref; ia.stall = h->pars.ipm_stall;
        ia.feastol = h->pars.ipm_feastol; ia.abstol = h->pars.ipm_abstol; ia.reltol = h->pars.ipm_reltol; ia.reg = h->pars.ipm_reg; ia.ref_gap = h->pars.ipm_ref_gap; ia.ref_tol = h->pars.ipm_ref_tol; ia.split_step = h->pars.ipm_split_step;
        ia.slab = h->slab; ia.slab_stride = h->slab_stride; ia.work = h->work; ia.work_stride = h->work_stride;
        ia.z_out = h->z_out; ia.p_out = h->p_out; ia.status = h->ipm_status; ia.iters = h->ipm_iters; ia.info = h->ipm_info;
        ia.active = h->active; ia.prof = h->prof;
        // warm start only inside a running PTR loop, from the second iteration on (the workspace then holds the snapshots of the
        // previous subproblem's solve and h->dev the previous solution's deviation)
        ia.warm_allowed = (h->run_ready && h->iter >= 2 && h->pars.ipm_warm != 0) ? 1 : 0;
        ia.warm_min_cold = h->pars.ipm_warm_min_cold;
        {   // snapshot levels, coarse ... very fine (scp_ptr_params: <= 0 selects the default of a level)
            auto dflt = [](double v, double d) { return v > 0.0 ? v : d; };
            const scp_ptr_params& q = h->pars;
            ia.warm_mu[0] = dflt(q.ipm_warm_mu_coarse, 1e-1); ia.warm_dev[0] = std::numeric_limits<double>::infinity();
            ia.warm_mu[1] = dflt(q.ipm_warm_mu_mid, 1e-5);    ia.warm_dev[1] = dflt(q.ipm_warm_dev_mid, 1e-1);
            ia.warm_mu[2] = q.ipm_warm_mu;                    ia.warm_dev[2] = q.ipm_warm_dev;
            ia.warm_mu[3] = dflt(q.ipm_warm_mu_vfine, 1e-10); ia.warm_dev[3] = dflt(q.ipm_warm_dev_vfine, 1e-6);
        }
        ia.prev_dev = h->dev; ia.cold_iters = h->cold_iters; ia.snap = h->snap;
        ExtractArgs ea;
        ea.B = B; ea.N = h->N; ea.slab = h->slab; ea.slab_stride = h->slab_stride; ea.z = h->z_out; ea.ph = h->p_out;
        ea.Sx = h->d_Sx; ea.cx = h->d_cx; ea.Su = h->d_Su; ea.cu = h->d_cu; ea.Sp = h->d_Sp; ea.cp = h->d_cp;
        ea.active = h->active; ea.xd = h->sol_xd; ea.ud = h->sol_ud; ea.p = h->sol_p; ea.cost = h->cost; ea.dev = h->dev;
        ea.eta = h->eta;
        ea.Eref = h->ref_dyn.E; ea.vd = h->vd; ea.vs = h->vs; ea.vic = h->vic; ea.vtc = h->vtc; ea.Ppen = h->Ppen; ea.Pf = h->Pf;
        ea.wvc = h->pars.wvc;
        // K4a runs in the tail of the solving wave (ipm2_solve_kernel's second argument) unless SCP_K3_FUSE_EXTRACT=0 asks for
        // the separate launch (A/B measurements; the results are bit-identical: the same code on the same data)
        static const bool fuse = []() { const char* e = getenv("SCP_K3_FUSE_EXTRACT"); return !(e && e[0] == '0'); }();
        ExtractArgs ef = ea;
        if (!fuse) ef.xd = nullptr;
        TRY(stamp_begin(h, 2));
        {
#ifdef SCP_IPM_ONLY_WPE   // experiment: a library with a single kernel variant
            hipLaunchKernelGGL((ipm2_solve_kernel<M, SCP_IPM_ONLY_WPE>), dim3(B), dim3(64), 0, h->stream, ia, ef);
#else
            int wpe = (B > 4 * h->num_cus) ? 2 : 1;   // more problems than SIMDs: two problems per SIMD
            if (h->pars.ipm_wpe == 1 || h->pars.ipm_wpe == 2) wpe = h->pars.ipm_wpe;
            if (h->wpe_override > 0) wpe = h->wpe_override;
            if (wpe >= 2) hipLaunchKernelGGL((ipm2_solve_kernel<M, 2>), dim3(B), dim3(64), 0, h->stream, ia, ef);
            else hipLaunchKernelGGL((ipm2_solve_kernel<M, 1>), dim3(B), dim3(64), 0, h->stream, ia, ef);
#endif
        }
        TRY(stamp_end(h));
        HIP_TRY(h, hipGetLastError());
        if (!fuse) {
            TRY(stamp_begin(h, 3));
            hipLaunchKernelGGL(ptr_extract_kernel<M>, dim3(B), dim3(64), 0, h->stream, ea);
            TRY(stamp_end(h));
            HIP_TRY(h, hipGetLastError());
        }
        h->sub_ready = true;
        return (int)SCP_OK;
    });
}

static int set_active_all(scp_problem* h, int B)
{
    std::vector<int> ones(B, 1);
    HIP_TRY(h, hipMemcpyAsync(h->active, ones.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    stamps_collect(h);
    return SCP_OK;
}

static int ptr_start_dev(scp_problem* h)
{
    const int B = h->B;
    const size_t nx = h->info.nx, nu = h->info.nu, np = h->npt, N = h->N, D = sizeof(double), b = B;
    HIP_TRY(h, hipMemcpyAsync(h->ref_xd, h->guess_xd, nx * N * b * D, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->ref_ud, h->guess_ud, nu * N * b * D, hipMemcpyDeviceToDevice, h->stream));
    if (np > 0) HIP_TRY(h, hipMemcpyAsync(h->ref_p, h->guess_p, np * b * D, hipMemcpyDeviceToDevice, h->stream));
    // until the first iteration has run, the "solution" returned by scp_ptr_get_host is the guess itself
    HIP_TRY(h, hipMemcpyAsync(h->sol_xd, h->guess_xd, nx * N * b * D, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->sol_ud, h->guess_ud, nu * N * b * D, hipMemcpyDeviceToDevice, h->stream));
    if (np > 0) HIP_TRY(h, hipMemcpyAsync(h->sol_p, h->guess_p, np * b * D, hipMemcpyDeviceToDevice, h->stream));
    h->iter = 0;
    // generate_initial_guess: discretize!(guess)  (ptr.jl:548-555); J_aug of the guess is NaN (ptr.jl:350)
    TRY(discretize_dev(h, B, h->ref_xd, h->ref_ud, h->ref_p, h->ref_dyn, h->d_feas_new, nullptr));
    // scp_ptr_get_host straight after init / restart returns the guess: its feasibility flag and defects are the guess's
    HIP_TRY(h, hipMemcpyAsync(h->d_feas, h->d_feas_new, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->sol_dyn.defect, h->ref_dyn.defect, nx * (N - 1) * b * D, hipMemcpyDeviceToDevice, h->stream));
    std::vector<double> nan(B, std::numeric_limits<double>::quiet_NaN());
    HIP_TRY(h, hipMemcpyAsync(h->Jaug_ref, nan.data(), (size_t)B * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->scp_status, 0, (size_t)B * sizeof(int), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->iters_done, 0, (size_t)B * sizeof(int), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->cold_iters, 0, (size_t)B * sizeof(int), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->snap, 0, (size_t)B * sizeof(int), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->hist, 0, (size_t)h->pars.iter_max * B * H_N * sizeof(double), h->stream));
    TRY(set_active_all(h, B));
    return SCP_OK;
}

extern "C" int scp_ptr_init_host(scp_handle h, int B, const scp_ptr_params* pars, const double* xd, const double* ud,
                                 const double* p, const double* pp)
{
    if (!h || B < 1 || !xd || !ud) return SCP_ERR_BAD_ARGUMENT;
    if (B > h->cap) return SCP_ERR_BATCH_TOO_LARGE;
    if (h->npt > 0 && !p) return SCP_ERR_BAD_ARGUMENT;
    if (h->info.npp > 0 && !pp) return SCP_ERR_BAD_ARGUMENT;
    TRY(check_pars(pars));
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(ensure_ptr_buffers(h, pars->iter_max));
    h->pars = *pars; h->B = B; h->iter = 0; h->run_ready = true; h->sub_ready = false;
    TRY(upload_traj(h, B, xd, ud, p, h->guess_xd, h->guess_ud, h->guess_p));
    if (h->info.npp > 0)
        HIP_TRY(h, hipMemcpyAsync(h->d_pp, pp, (size_t)h->info.npp * B * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return ptr_start_dev(h);
}

extern "C" int scp_ptr_init_guess_host(scp_handle h, int B, const scp_ptr_params* pars, const double* pp)
{
    if (!h || B < 1) return SCP_ERR_BAD_ARGUMENT;
    if (B > h->cap) return SCP_ERR_BATCH_TOO_LARGE;
    if (h->info.npp > 0 && !pp) return SCP_ERR_BAD_ARGUMENT;
    TRY(check_pars(pars));
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(ensure_ptr_buffers(h, pars->iter_max));
    h->pars = *pars; h->B = B; h->iter = 0; h->run_ready = true; h->sub_ready = false;
    if (h->info.npp > 0)
        HIP_TRY(h, hipMemcpyAsync(h->d_pp, pp, (size_t)h->info.npp * B * sizeof(double), hipMemcpyHostToDevice, h->stream));
    GuessArgs g;
    g.B = B; g.N = h->N; g.pp = h->d_pp; g.xd = h->guess_xd; g.ud = h->guess_ud; g.p = h->guess_p;
    TRY(with_model(h->model_id, [&](auto m) -> int {
        using M = decltype(m);
        typename M::Params P = M::make_params(h->par.data());
        const long n = (long)B * h->N;
        hipLaunchKernelGGL(ptr_guess_kernel<M>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, g, P);
        return (int)SCP_OK;
    }));
    HIP_TRY(h, hipGetLastError());
    return ptr_start_dev(h);
}

// ---- the reference's Starship guess on the device (starship_guess.hpp): flip simulation -> batched descent programs -> reconstruction ----
struct StarshipGuessState {
    scp::conic::Engine eng;
    scp::SgPattern pat;
    int chunk = 0;                       // instances per conic launch
    std::vector<void*> allocs;
    int *a_kind = nullptr, *a_i = nullptr, *a_j = nullptr, *g_kind = nullptr, *b_kind = nullptr, *b_i = nullptr, *h_kind = nullptr;
    int *a_row = nullptr, *g_row = nullptr, *gs_kind = nullptr;
    double *g_val = nullptr, *h_val = nullptr, *lti = nullptr, *gs_val = nullptr;
    double *xs = nullptr, *t1 = nullptr;
    int *ok1 = nullptr, *active = nullptr, *fail = nullptr;
    double Su[2], cu[2];
    int n1 = 0, N2 = 0, id_sw = 0;
};
static void starship_guess_free(StarshipGuessState* g)
{
    if (!g) return;
    g->eng.destroy();
    for (void* p : g->allocs) (void)hipFree(p);
    delete g;
}
template <class T>
static int sg_upload(scp_problem* h, StarshipGuessState* g, T** dst, const std::vector<T>& v)
{
    void* d = nullptr;
    HIP_TRY(h, hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    g->allocs.push_back(d);
    if (!v.empty()) HIP_TRY(h, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = (T*)d;
    return SCP_OK;
}
static int starship_guess_dev(scp_problem* h, int B, const double* d_pp, double* d_xd, double* d_ud, double* d_p)
{
    using namespace scp;
    const Starship::Params K = Starship::make_params(h->par.data());
    const int N = h->N;
    if (!h->sg) {
        // built into a local object and published in h->sg only after EVERY step succeeded: a half-built state (chunk = 0, null
        // device arrays, engine not created) must never be seen by the next call on this handle
        StarshipGuessState* g = new (std::nothrow) StarshipGuessState;
        if (!g) return SCP_ERR_ALLOC;
        struct Guard { StarshipGuessState* g; ~Guard() { if (g) starship_guess_free(g); } } guard{g};
        // grid split (definition.jl:108-113): id1 = {k: tau_k <= tau_s}, id2 = id1[end] .. N
        int n1 = 0;
        for (int k = 0; k < N; k++) { const double t = (double)k / (double)(N - 1); if ((1.0 - t) * 0.0 + t * 1.0 <= K.tau_s) n1 = k + 1; }
        if (n1 < 2 || n1 >= N) { h->err = "starship guess: the grid has no node on both sides of tau_s"; return SCP_ERR_BAD_ARGUMENT; }
        g->n1 = n1; g->id_sw = n1 - 1; g->N2 = N - g->id_sw;
        const double Tmax_x = K.T_max1 * std::sin(K.theta_max2);
        sg_scale(-Tmax_x, Tmax_x, g->Su[0], g->cu[0]); sg_scale(K.T_min1, K.T_max1, g->Su[1], g->cu[1]);
        g->pat = sg_build_pattern(g->N2, g->Su, g->cu, K.T_min1, K.T_max1, K.theta_max2);
        // FOH models of the candidate durations: one normalised interval of the phase-2 grid
        auto tau = [&](int k) { const double t = (double)k / (double)(N - 1); return (1.0 - t) * 0.0 + t * 1.0; };
        const double dtn = (tau(g->id_sw + 1) - tau(g->id_sw)) - (tau(g->id_sw) - tau(g->id_sw));
        std::vector<double> lti((size_t)SG_NCAND * 36);
        for (int c = 0; c < SG_NCAND; c++) {
            double o[36];
            sg_descent_lti(dtn, (10.0 + c) / (1.0 - K.tau_s), K.m, K.g0, o);
            std::copy(o, o + 36, lti.begin() + (size_t)c * 36);
        }
        g->chunk = std::min(h->cap, 256);
        conic::Csc Pm; Pm.nrow = g->pat.n; Pm.ncol = g->pat.n; Pm.p.assign(g->pat.n + 1, 0);
        int rc = g->eng.create(g->pat.n, g->pat.p, g->pat.m, g->pat.l, g->pat.q, Pm, g->pat.A, g->pat.G, nullptr, g->chunk * SG_NCAND, h->device);
        if (rc != SCP_OK) { h->err = "starship guess: " + g->eng.err; return rc; }
        TRY(sg_upload(h, g, &g->a_kind, g->pat.a_kind)); TRY(sg_upload(h, g, &g->a_i, g->pat.a_i)); TRY(sg_upload(h, g, &g->a_j, g->pat.a_j));
        TRY(sg_upload(h, g, &g->g_kind, g->pat.g_kind)); TRY(sg_upload(h, g, &g->g_val, g->pat.g_val));
        TRY(sg_upload(h, g, &g->b_kind, g->pat.b_kind)); TRY(sg_upload(h, g, &g->b_i, g->pat.b_i));
        TRY(sg_upload(h, g, &g->h_kind, g->pat.h_kind)); TRY(sg_upload(h, g, &g->h_val, g->pat.h_val));
        TRY(sg_upload(h, g, &g->lti, lti));
        TRY(sg_upload(h, g, &g->a_row, g->pat.A.i)); TRY(sg_upload(h, g, &g->g_row, g->pat.G.i));
        TRY(sg_upload(h, g, &g->gs_kind, g->pat.gs_kind)); TRY(sg_upload(h, g, &g->gs_val, g->pat.gs_val));
        TRY(sg_upload(h, g, &g->xs, std::vector<double>((size_t)8 * h->cap, 0.0))); TRY(sg_upload(h, g, &g->t1, std::vector<double>((size_t)h->cap, 0.0)));
        TRY(sg_upload(h, g, &g->ok1, std::vector<int>((size_t)h->cap, 0))); TRY(sg_upload(h, g, &g->fail, std::vector<int>((size_t)h->cap, 0)));
        TRY(sg_upload(h, g, &g->active, std::vector<int>((size_t)g->chunk * SG_NCAND, 0)));
        if (g->chunk <= 0) { h->err = "starship guess: empty batch capacity"; return SCP_ERR_BAD_ARGUMENT; }
        h->sg = g;
        guard.g = nullptr;
    }
    StarshipGuessState* g = h->sg;
    SgDev a;
    a.B = B; a.N = N; a.n1 = g->n1; a.N2 = g->N2; a.id_sw = g->id_sw; a.pp = d_pp; a.xd = d_xd; a.ud = d_ud; a.p = d_p;
    a.xs = g->xs; a.t1 = g->t1; a.ok1 = g->ok1;
    hipLaunchKernelGGL(starship_flip_kernel, dim3((B + 63) / 64), dim3(64), 0, h->stream, a, K);
    HIP_TRY(h, hipGetLastError());
    SgProg P;
    P.n = g->pat.n; P.p = g->pat.p; P.m = g->pat.m; P.l = g->pat.l; P.nnzA = g->pat.A.nnz(); P.nnzG = g->pat.G.nnz(); P.N2 = g->N2;
    P.a_kind = g->a_kind; P.a_i = g->a_i; P.a_j = g->a_j; P.g_kind = g->g_kind; P.g_val = g->g_val; P.b_kind = g->b_kind; P.b_i = g->b_i;
    P.h_kind = g->h_kind; P.h_val = g->h_val; P.lti = g->lti;
    P.a_row = g->a_row; P.g_row = g->g_row; P.gs_kind = g->gs_kind; P.gs_val = g->gs_val;
    P.Su[0] = g->Su[0]; P.Su[1] = g->Su[1]; P.cu[0] = g->cu[0]; P.cu[1] = g->cu[1]; P.vf[0] = K.vf_x; P.vf[1] = K.vf_y;
    conic::Opts o = conic::default_opts();
    o.nref = 30;      // feasibility programs (zero cost, variables held by equality rows only) need more refinement steps (models.py)
    for (int b0 = 0; b0 < B; b0 += g->chunk) {
        const int nb = std::min(g->chunk, B - b0);
        SgFill f;
        f.B = nb; f.BS = g->eng.BS; f.xs = g->xs + (size_t)8 * b0; f.ok1 = g->ok1 + b0;
        f.c = g->eng.c; f.b = g->eng.b; f.h = g->eng.h; f.Gx = g->eng.Gx; f.Ax = g->eng.Ax; f.active = g->active;
        const long nt = (long)nb * SG_NCAND;
        hipLaunchKernelGGL(starship_descent_fill_kernel, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, h->stream, f, P);
        HIP_TRY(h, hipGetLastError());
        int rc = g->eng.launch(h->stream, (int)nt, o, 0u, g->active);
        if (rc != SCP_OK) { h->err = "starship guess: " + g->eng.err; return rc; }
        SgRec r;
        r.B = nb; r.N = N; r.n1 = g->n1; r.N2 = g->N2; r.id_sw = g->id_sw; r.BS = g->eng.BS; r.z = g->eng.x; r.status = g->eng.status;
        r.xs = g->xs + (size_t)8 * b0; r.t1 = g->t1 + b0; r.ok1 = g->ok1 + b0;
        r.xd = d_xd + (size_t)b0 * N * 8; r.ud = d_ud + (size_t)b0 * N * 3; r.p = d_p + (size_t)b0 * 10; r.fail = g->fail + b0;
        r.Su[0] = g->Su[0]; r.Su[1] = g->Su[1]; r.cu[0] = g->cu[0]; r.cu[1] = g->cu[1]; r.tau_s = K.tau_s; r.alpha_e = K.alpha_e;
        hipLaunchKernelGGL(starship_reconstruct_kernel, dim3((nb + 63) / 64), dim3(64), 0, h->stream, r);
        HIP_TRY(h, hipGetLastError());
    }
    // instances without a reference guess (no velocity crossing / no feasible descent duration: the reference raises an error,
    // definition.jl:163-167, 415-419) get the straight-line guess and are counted (scp_guess_failures)
    std::vector<int> fail(B);
    HIP_TRY(h, hipMemcpyAsync(fail.data(), g->fail, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    int nf = 0;
    for (int b = 0; b < B; b++) nf += fail[b] != 0;
    h->guess_failures = nf;
    if (nf > 0) {
        GuessArgs ga;
        ga.B = B; ga.N = N; ga.pp = d_pp; ga.xd = d_xd; ga.ud = d_ud; ga.p = d_p; ga.only = g->fail;
        hipLaunchKernelGGL(ptr_guess_kernel<Starship>, dim3((unsigned)(((long)B * N + 255) / 256)), dim3(256), 0, h->stream, ga, K);
        HIP_TRY(h, hipGetLastError());
    }
    return SCP_OK;
}

extern "C" int scp_guess_failures(scp_handle h) { return h ? h->guess_failures : -1; }

// traj.guess(N) of the compiled model for a Monte-Carlo batch, evaluated on the device for ANY registered model (the
// structured ones also have scp_ptr_init_guess_host, which keeps the guesses resident for a PTR run)
extern "C" int scp_guess_batch_host(scp_handle h, int B, const double* pp, double* xd, double* ud, double* p)
{
    if (!h || B < 1 || !xd || !ud) return SCP_ERR_BAD_ARGUMENT;
    if (B > h->cap) return SCP_ERR_BATCH_TOO_LARGE;
    if ((h->info.npp > 0 && !pp) || (h->npt > 0 && !p)) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    // a pure query: its own scratch, so that a resident PTR / SCvx / GuSTO run (d_pp, sol_*) is left untouched
    const size_t nx = h->info.nx, nu = h->info.nu, np = h->npt, npp = h->info.npp, N = h->N, D = sizeof(double), b = B;
    if (!h->q_pp) {
        TRY(dalloc(h, &h->q_pp, (npp > 0 ? npp : 1) * (size_t)h->cap)); TRY(dalloc(h, &h->q_xd, nx * N * h->cap));
        TRY(dalloc(h, &h->q_ud, nu * N * h->cap)); TRY(dalloc(h, &h->q_p, (np > 0 ? np : 1) * (size_t)h->cap));
    }
    if (npp > 0) HIP_TRY(h, hipMemcpyAsync(h->q_pp, pp, npp * b * D, hipMemcpyHostToDevice, h->stream));
    GuessArgs g;
    g.B = B; g.N = h->N; g.pp = h->q_pp; g.xd = h->q_xd; g.ud = h->q_ud; g.p = h->q_p;
    h->guess_failures = 0;
    if (h->model_id == Starship::id && !std::getenv("SCP_STARSHIP_STRAIGHT_LINE_GUESS")) {
        // the reference's own guess: bang-bang flip + convex terminal descent, per instance (starship_guess.hpp)
        TRY(starship_guess_dev(h, B, h->q_pp, h->q_xd, h->q_ud, h->q_p));
    } else {
    TRY(with_model(h->model_id, [&](auto m) -> int {
        using M = decltype(m);
        typename M::Params P = M::make_params(h->par.data());
        const long n = (long)B * h->N;
        hipLaunchKernelGGL(ptr_guess_kernel<M>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, g, P);
        return (int)SCP_OK;
    }));
    }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(xd, h->q_xd, nx * N * b * D, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(ud, h->q_ud, nu * N * b * D, hipMemcpyDeviceToHost, h->stream));
    if (np > 0) HIP_TRY(h, hipMemcpyAsync(p, h->q_p, np * b * D, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SCP_OK;
}

extern "C" int scp_ptr_restart(scp_handle h)
{
    if (!h || !h->run_ready || h->B < 1 || h->pars.iter_max > h->hist_cap) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    return ptr_start_dev(h);
}

extern "C" int scp_get_kernel_timing(scp_handle h, double seconds[4], long launches[4], int reset)
{
    if (!h) return SCP_ERR_BAD_ARGUMENT;
    for (int i = 0; i < 4; i++) {
        if (seconds) seconds[i] = h->t_kernel[i];
        if (launches) launches[i] = h->n_kernel[i];
        if (reset) { h->t_kernel[i] = 0; h->n_kernel[i] = 0; }
    }
    return SCP_OK;
}

static int copy_sol_to_ref(scp_problem* h, int B)
{
    const size_t nx = h->info.nx, nu = h->info.nu, np = h->npt, npF = h->info.npF > 0 ? h->info.npF : 1, N = h->N,
                 M = N - 1, D = sizeof(double), b = B;
    auto cp = [&](double* dst, const double* src, size_t n) { return hipMemcpyAsync(dst, src, n * D, hipMemcpyDeviceToDevice, h->stream); };
    HIP_TRY(h, cp(h->ref_xd, h->sol_xd, nx * N * b)); HIP_TRY(h, cp(h->ref_ud, h->sol_ud, nu * N * b));
    if (np > 0) HIP_TRY(h, cp(h->ref_p, h->sol_p, np * b));
    HIP_TRY(h, cp(h->ref_dyn.A, h->sol_dyn.A, nx * nx * M * b)); HIP_TRY(h, cp(h->ref_dyn.Bm, h->sol_dyn.Bm, nx * nu * M * b));
    HIP_TRY(h, cp(h->ref_dyn.Bp, h->sol_dyn.Bp, nx * nu * M * b)); HIP_TRY(h, cp(h->ref_dyn.F, h->sol_dyn.F, nx * npF * M * b));
    HIP_TRY(h, cp(h->ref_dyn.r, h->sol_dyn.r, nx * M * b));
    HIP_TRY(h, cp(h->ref_dyn.E, h->sol_dyn.E, nx * nx * M * b));   // ref.dyn.E enters the next subproblem's vd (ptr.jl:805)
    return SCP_OK;
}

__global__ void merge_feas_kernel(int B, const int* active, const int* fnew, int* feas)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && (active == nullptr || active[b])) feas[b] = fnew[b];
}

// Enqueues one PTR iteration on the handle's stream WITHOUT waiting for it: several handles (sub-batches, one stream each)
// then overlap on the GPU, and several iterations can be in flight per handle -- the straggling problems of one launch no
// longer idle the rest of the chip.  scp_ptr_poll waits and returns the active count.
extern "C" int scp_ptr_iterate_async(scp_handle h)
{
    if (!h || !h->run_ready || h->B < 1 || h->pars.iter_max > h->hist_cap) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const int B = h->B;
    h->iter += 1;
    if (h->iter > h->pars.iter_max) return SCP_OK;
    TRY(subproblem_dev(h, B));
    // SCPSubproblemSolution(spbm, ctor) -> SubproblemSolution(x,u,p,...) -> discretize! (ptr.jl:380)
    TRY(discretize_dev(h, B, h->sol_xd, h->sol_ud, h->sol_p, h->sol_dyn, h->d_feas_new, h->active));
    hipLaunchKernelGGL(merge_feas_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, B, h->active, h->d_feas_new, h->d_feas);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemsetAsync(h->n_active, 0, sizeof(int), h->stream));
    UpdateArgs ua;
    ua.B = B; ua.iter = h->iter; ua.iter_max = h->pars.iter_max; ua.eps_abs = h->pars.eps_abs; ua.eps_rel = h->pars.eps_rel;
    ua.cost = h->cost; ua.dev = h->dev; ua.feas = h->d_feas; ua.ipm_status = h->ipm_status; ua.ipm_iters = h->ipm_iters;
    ua.ipm_info = h->ipm_info; ua.Jaug_ref = h->Jaug_ref; ua.active = h->active; ua.scp_status = h->scp_status;
    ua.iters_done = h->iters_done; ua.hist = h->hist; ua.n_active = h->n_active;
    TRY(stamp_begin(h, 3));
    hipLaunchKernelGGL(ptr_update_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, ua);
    TRY(stamp_end(h));
    HIP_TRY(h, hipGetLastError());
    // the active count of THIS iteration, readable later without draining the stream (scp_ptr_poll_iteration)
    if (h->na_cap < h->pars.iter_max + 2) {     // (first iteration of a run with a longer horizon: nothing of the ring is in flight)
        if (h->na_ring) { HIP_TRY(h, hipStreamSynchronize(h->stream)); HIP_TRY(h, hipHostFree(h->na_ring)); h->na_ring = nullptr; }
        if (h->na_dev) { HIP_TRY(h, hipFree(h->na_dev)); h->na_dev = nullptr; }
        h->na_cap = h->pars.iter_max + 2;
        HIP_TRY(h, hipHostMalloc((void**)&h->na_ring, sizeof(int) * (size_t)h->na_cap));
        HIP_TRY(h, hipMalloc((void**)&h->na_dev, sizeof(int) * (size_t)h->na_cap));
    }
    while ((int)h->na_ev.size() <= h->iter) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->na_ev.push_back(e);
    }
    HIP_TRY(h, hipMemcpyAsync(&h->na_dev[h->iter], h->n_active, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&h->na_ring[h->iter], h->n_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipEventRecord(h->na_ev[h->iter], h->stream));
    // ref = spbm.sol (ptr.jl:509).  Whole-batch copy: problems that stopped are never read again as `ref`.
    TRY(copy_sol_to_ref(h, B));
    return SCP_OK;
}

extern "C" int scp_ptr_poll(scp_handle h, int* n_active)
{
    if (!h || !h->run_ready || h->B < 1) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    int na = 0;
    if (h->iter >= 1 && h->iter <= h->pars.iter_max)   // n_active of the last enqueued iteration (0 once iter_max is passed)
        HIP_TRY(h, hipMemcpyAsync(&na, h->n_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    stamps_collect(h);
    if (n_active) *n_active = na;
    return SCP_OK;
}

// Active count at the end of iteration `iteration` (1-based, already enqueued) WITHOUT waiting for later work on the stream: the
// caller enqueues window k + 1, then reads the count of window k (multi-GPU loop: the queue never drains at a window boundary).
extern "C" int scp_ptr_poll_iteration(scp_handle h, int iteration, int* n_active)
{
    if (!h || !h->run_ready || h->B < 1 || !n_active || iteration < 1 || iteration > h->iter) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    if (iteration > h->pars.iter_max) { *n_active = 0; return SCP_OK; }     // nothing was enqueued beyond iter_max
    if (!h->na_ring || (int)h->na_ev.size() <= iteration) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipEventSynchronize(h->na_ev[iteration]));
    *n_active = h->na_ring[iteration];
    stamps_collect_ready(h);      // fold the kernel time stamps that have completed (without waiting) -- the pending list stays short
    return SCP_OK;
}

extern "C" int scp_ptr_iterate(scp_handle h, int* n_active)
{
    TRY(scp_ptr_iterate_async(h));
    return scp_ptr_poll(h, n_active);
}

extern "C" int scp_ptr_get_host(scp_handle h, double* xd, double* ud, double* p, int32_t* status, int32_t* iterations,
                                double* cost, uint8_t* feas, double* defect, double* hist)
{
    if (!h || !h->run_ready || h->B < 1 || h->pars.iter_max > h->hist_cap) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nx = h->info.nx, nu = h->info.nu, np = h->npt, N = h->N, D = sizeof(double), b = h->B;
    if (xd) HIP_TRY(h, hipMemcpyAsync(xd, h->sol_xd, nx * N * b * D, hipMemcpyDeviceToHost, h->stream));
    if (ud) HIP_TRY(h, hipMemcpyAsync(ud, h->sol_ud, nu * N * b * D, hipMemcpyDeviceToHost, h->stream));
    if (p && np > 0) HIP_TRY(h, hipMemcpyAsync(p, h->sol_p, np * b * D, hipMemcpyDeviceToHost, h->stream));
    if (status) HIP_TRY(h, hipMemcpyAsync(status, h->scp_status, b * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (iterations) HIP_TRY(h, hipMemcpyAsync(iterations, h->iters_done, b * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (cost) HIP_TRY(h, hipMemcpyAsync(cost, h->cost, 4 * b * D, hipMemcpyDeviceToHost, h->stream));
    if (defect) HIP_TRY(h, hipMemcpyAsync(defect, h->sol_dyn.defect, nx * (N - 1) * b * D, hipMemcpyDeviceToHost, h->stream));
    if (hist) HIP_TRY(h, hipMemcpyAsync(hist, h->hist, (size_t)h->pars.iter_max * b * H_N * D, hipMemcpyDeviceToHost, h->stream));
    TRY(feas_out(h, h->B, h->d_feas, feas));
    return SCP_OK;
}

extern "C" int scp_ptr_solve_batch_host(scp_handle h, int B, const scp_ptr_params* pars, const double* xd,
                                        const double* ud, const double* p, const double* pp, double* xd_out,
                                        double* ud_out, double* p_out, int32_t* status, int32_t* iterations,
                                        double* cost, uint8_t* feas, double* seconds)
{
    if (!h) return SCP_ERR_BAD_ARGUMENT;
    TRY(scp_ptr_init_host(h, B, pars, xd, ud, p, pp));
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    int na = B;
    while (na > 0) TRY(scp_ptr_iterate(h, &na));
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    TRY(scp_ptr_get_host(h, xd_out, ud_out, p_out, status, iterations, cost, feas, nullptr, nullptr));
    if (seconds) {
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        *seconds = ms * 1e-3;
    }
    return SCP_OK;
}

extern "C" int scp_ptr_solve_subproblem_batch_host(scp_handle h, int B, const scp_ptr_params* pars,
                                                   const double* xd_ref, const double* ud_ref, const double* p_ref,
                                                   const double* pp, double* x, double* u, double* p, double* cost,
                                                   double* eta, int32_t* solver_status, int32_t* solver_iters,
                                                   double* info, double* defect, uint8_t* feas, double* seconds)
{
    if (!h || B < 1 || !xd_ref || !ud_ref) return SCP_ERR_BAD_ARGUMENT;
    if (B > h->cap) return SCP_ERR_BATCH_TOO_LARGE;
    if (h->npt > 0 && !p_ref) return SCP_ERR_BAD_ARGUMENT;
    if (h->info.npp > 0 && !pp) return SCP_ERR_BAD_ARGUMENT;
    TRY(check_pars(pars));
    HIP_TRY(h, hipSetDevice(h->device));
    TRY(ensure_ptr_buffers(h, 1));
    // a stand-alone subproblem solve reuses the run's trajectory buffers: any initialised PTR run ends here
    // (restart / iterate / get_host are refused until the next scp_ptr_init_*)
    h->pars = *pars; h->B = B; h->iter = 0; h->run_ready = false;
    TRY(upload_traj(h, B, xd_ref, ud_ref, p_ref, h->ref_xd, h->ref_ud, h->ref_p));
    if (h->info.npp > 0)
        HIP_TRY(h, hipMemcpyAsync(h->d_pp, pp, (size_t)h->info.npp * B * sizeof(double), hipMemcpyHostToDevice, h->stream));
    TRY(set_active_all(h, B));
    TRY(discretize_dev(h, B, h->ref_xd, h->ref_ud, h->ref_p, h->ref_dyn, h->d_feas_new, nullptr));
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    TRY(subproblem_dev(h, B));
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    TRY(discretize_dev(h, B, h->sol_xd, h->sol_ud, h->sol_p, h->sol_dyn, h->d_feas_new, nullptr));
    const size_t nx = h->info.nx, nu = h->info.nu, np = h->npt, N = h->N, D = sizeof(double), b = B;
    if (x) HIP_TRY(h, hipMemcpyAsync(x, h->sol_xd, nx * N * b * D, hipMemcpyDeviceToHost, h->stream));
    if (u) HIP_TRY(h, hipMemcpyAsync(u, h->sol_ud, nu * N * b * D, hipMemcpyDeviceToHost, h->stream));
    if (p && np > 0) HIP_TRY(h, hipMemcpyAsync(p, h->sol_p, np * b * D, hipMemcpyDeviceToHost, h->stream));
    if (cost) HIP_TRY(h, hipMemcpyAsync(cost, h->cost, 4 * b * D, hipMemcpyDeviceToHost, h->stream));
    if (eta) HIP_TRY(h, hipMemcpyAsync(eta, h->eta, (2 * N + 1) * b * D, hipMemcpyDeviceToHost, h->stream));
    if (solver_status) HIP_TRY(h, hipMemcpyAsync(solver_status, h->ipm_status, b * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (solver_iters) HIP_TRY(h, hipMemcpyAsync(solver_iters, h->ipm_iters, b * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (info) HIP_TRY(h, hipMemcpyAsync(info, h->ipm_info, 8 * b * D, hipMemcpyDeviceToHost, h->stream));
    if (defect) HIP_TRY(h, hipMemcpyAsync(defect, h->sol_dyn.defect, nx * (N - 1) * b * D, hipMemcpyDeviceToHost, h->stream));
    TRY(feas_out(h, B, h->d_feas_new, feas));
    if (seconds) {
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        *seconds = ms * 1e-3;
    }
    return SCP_OK;
}

extern "C" int scp_ptr_get_virtual_controls_host(scp_handle h, double* vd, double* vs, double* vic, double* vtc, double* P,
                                                 double* Pf)
{
    if (!h || !h->ptr_ready || !h->sub_ready || h->B < 1) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nx = h->info.nx, ns = h->info.ns, nic = h->info.nic, ntc = h->info.ntc, N = h->N, D = sizeof(double), b = h->B;
    if (vd) HIP_TRY(h, hipMemcpyAsync(vd, h->vd, nx * (N - 1) * b * D, hipMemcpyDeviceToHost, h->stream));
    if (vs && ns > 0) HIP_TRY(h, hipMemcpyAsync(vs, h->vs, ns * N * b * D, hipMemcpyDeviceToHost, h->stream));
    if (vic && nic > 0) HIP_TRY(h, hipMemcpyAsync(vic, h->vic, nic * b * D, hipMemcpyDeviceToHost, h->stream));
    if (vtc && ntc > 0) HIP_TRY(h, hipMemcpyAsync(vtc, h->vtc, ntc * b * D, hipMemcpyDeviceToHost, h->stream));
    if (P) HIP_TRY(h, hipMemcpyAsync(P, h->Ppen, N * b * D, hipMemcpyDeviceToHost, h->stream));
    if (Pf) HIP_TRY(h, hipMemcpyAsync(Pf, h->Pf, 2 * b * D, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    stamps_collect(h);
    return SCP_OK;
}

extern "C" int scp_debug_get_ipm_profile(scp_handle h, int b, long long* ticks8)
{
    if (!h || !h->ptr_ready || b < 0 || b >= h->cap || !ticks8) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(ticks8, h->prof + (long)b * 8, 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return SCP_OK;
}

extern "C" int scp_debug_get_stage_problem(scp_handle h, int b, double* buf, long* n_doubles)
{
    if (!h || !h->ptr_ready || b < 0 || b >= h->cap) return SCP_ERR_BAD_ARGUMENT;
    if (n_doubles) *n_doubles = h->slab_stride;
    if (buf) {
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipMemcpy(buf, h->slab + (long)b * h->slab_stride, (size_t)h->slab_stride * sizeof(double), hipMemcpyDeviceToHost));
    }
    return SCP_OK;
}

#include "scp_generic.hpp"


// =====================================================================================================================
// Multi-GPU behind the boundary (include/scp_mi355x.h, "Multi-GPU"): RCCL all-reduce of the device-resident active count.
// librccl.so is loaded lazily (dlopen) so that single-GPU callers never pay for it and the library has no link-time dependency.
// =====================================================================================================================
#include <dlfcn.h>
#include <rccl/rccl.h>

struct scp_comm {
    void* dl = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;            // the collective's stream (high priority: one tiny kernel must not queue behind K3 waves)
    long long *d_send = nullptr, *d_recv = nullptr, *h_ring = nullptr;
    int ring_cap = 0;
    std::vector<hipEvent_t> ev;              // ev[w]: the global count of window w has landed in h_ring[w]
    std::vector<hipEvent_t> part_ev;         // scratch events recorded on the parts' streams
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
static std::string g_comm_err;     // errors before a communicator exists (scp_comm_last_error(NULL))

static void* rccl_open(std::string& err)
{
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        if (void* d = dlopen(n, RTLD_NOW | RTLD_LOCAL)) return d;
    }
    err = std::string("dlopen(librccl.so): ") + (dlerror() ? dlerror() : "not found");
    return nullptr;
}

extern "C" const char* scp_comm_last_error(scp_comm_handle c) { return c ? c->err.c_str() : g_comm_err.c_str(); }

extern "C" void scp_shard_range(long n_total, int rank, int world, long* lo, long* hi)
{
    if (world < 1) world = 1;
    const long base = n_total / world, rem = n_total % world;
    const long l = (long)rank * base + std::min<long>(rank, rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

extern "C" int scp_comm_unique_id(unsigned char id[SCP_COMM_ID_BYTES])
{
    static_assert(SCP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id) return SCP_ERR_BAD_ARGUMENT;
    void* dl = rccl_open(g_comm_err);
    if (!dl) return SCP_ERR_UNSUPPORTED;
    auto get = (ncclResult_t(*)(ncclUniqueId*))dlsym(dl, "ncclGetUniqueId");
    if (!get) { g_comm_err = "ncclGetUniqueId not found"; return SCP_ERR_UNSUPPORTED; }
    ncclUniqueId u;
    const ncclResult_t r = get(&u);
    if (r != ncclSuccess) { g_comm_err = "ncclGetUniqueId failed"; return SCP_ERR_HIP; }
    std::memcpy(id, u.internal, SCP_COMM_ID_BYTES);
    return SCP_OK;     // (the library handle stays open: RCCL keeps bootstrap state behind the id)
}

extern "C" int scp_comm_preflight(int device)
{
    std::string err;
    void* dl = rccl_open(err);
    if (!dl) { g_comm_err = err; return SCP_ERR_UNSUPPORTED; }
    dlclose(dl);      // (only the check: scp_comm_create opens its own handle)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_comm_err = "scp_comm_preflight: no such device"; return SCP_ERR_NO_DEVICE; }
    hipStream_t st = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { g_comm_err = "scp_comm_preflight: device / stream set-up failed"; return SCP_ERR_HIP; }
    (void)hipStreamDestroy(st);
    return SCP_OK;
}

extern "C" void scp_comm_destroy(scp_comm_handle c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && c->CommDestroy) (void)c->CommDestroy(c->comm);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->part_ev) (void)hipEventDestroy(e);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->h_ring) (void)hipHostFree(c->h_ring);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

#define COMM_HIP(c, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) { (c)->err = std::string(#call) + ": " + hipGetErrorString(e_); return SCP_ERR_HIP; } \
    } while (0)

// world == 1 and id == NULL: a communicator without RCCL (the single-process form of the same loop)
extern "C" int scp_comm_create(const unsigned char id[SCP_COMM_ID_BYTES], int rank, int world, int device, scp_comm_handle* out)
{
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return SCP_ERR_BAD_ARGUMENT;
    scp_comm* c = new (std::nothrow) scp_comm;
    if (!c) return SCP_ERR_ALLOC;
    struct Guard { scp_comm* c; ~Guard() { if (c) { g_comm_err = c->err; scp_comm_destroy(c); } } } guard{c};
    c->rank = rank; c->world = world; c->device = device;
    COMM_HIP(c, hipSetDevice(device));
    int lo = 0, hi = 0;
    COMM_HIP(c, hipDeviceGetStreamPriorityRange(&lo, &hi));      // (numerically lowest = highest priority)
    COMM_HIP(c, hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
    if (id) {
        c->dl = rccl_open(c->err);
        if (!c->dl) return SCP_ERR_UNSUPPORTED;
        auto init = (ncclResult_t(*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(c->dl, "ncclCommInitRank");
        c->AllReduce = (decltype(c->AllReduce))dlsym(c->dl, "ncclAllReduce");
        c->CommDestroy = (decltype(c->CommDestroy))dlsym(c->dl, "ncclCommDestroy");
        c->GetErrorString = (decltype(c->GetErrorString))dlsym(c->dl, "ncclGetErrorString");
        if (!init || !c->AllReduce || !c->CommDestroy) { c->err = "RCCL entry points not found"; return SCP_ERR_UNSUPPORTED; }
        ncclUniqueId u;
        std::memcpy(u.internal, id, SCP_COMM_ID_BYTES);
        const ncclResult_t r = init(&c->comm, world, u, rank);
        if (r != ncclSuccess) { c->err = std::string("ncclCommInitRank: ") + (c->GetErrorString ? c->GetErrorString(r) : "failed"); c->comm = nullptr; return SCP_ERR_HIP; }
    }
    *out = c;
    guard.c = nullptr;
    return SCP_OK;
}

static int comm_ring(scp_comm* c, int windows)
{
    if (c->ring_cap >= windows) return SCP_OK;
    COMM_HIP(c, hipStreamSynchronize(c->stream));
    if (c->d_send) { (void)hipFree(c->d_send); c->d_send = nullptr; }
    if (c->d_recv) { (void)hipFree(c->d_recv); c->d_recv = nullptr; }
    if (c->h_ring) { (void)hipHostFree(c->h_ring); c->h_ring = nullptr; }
    COMM_HIP(c, hipMalloc((void**)&c->d_send, sizeof(long long) * (size_t)windows));
    COMM_HIP(c, hipMalloc((void**)&c->d_recv, sizeof(long long) * (size_t)windows));
    COMM_HIP(c, hipHostMalloc((void**)&c->h_ring, sizeof(long long) * (size_t)windows));
    while ((int)c->ev.size() < windows) { hipEvent_t e; COMM_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev.push_back(e); }
    c->ring_cap = windows;
    return SCP_OK;
}

// enqueue on the collective stream: recv[w] = SUM over ranks of send[w] -> pinned ring, event
static int comm_reduce_window(scp_comm* c, int w)
{
    if (c->comm) {
        const ncclResult_t r = c->AllReduce(c->d_send + w, c->d_recv + w, 1, ncclInt64, ncclSum, c->comm, c->stream);
        if (r != ncclSuccess) { c->err = std::string("ncclAllReduce: ") + (c->GetErrorString ? c->GetErrorString(r) : "failed"); return SCP_ERR_HIP; }
    } else {
        COMM_HIP(c, hipMemcpyAsync(c->d_recv + w, c->d_send + w, sizeof(long long), hipMemcpyDeviceToDevice, c->stream));
    }
    COMM_HIP(c, hipMemcpyAsync(c->h_ring + w, c->d_recv + w, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    COMM_HIP(c, hipEventRecord(c->ev[w], c->stream));
    return SCP_OK;
}

extern "C" int scp_comm_all_reduce_sum_i64(scp_comm_handle c, long long* value)
{
    if (!c || !value) return SCP_ERR_BAD_ARGUMENT;
    COMM_HIP(c, hipSetDevice(c->device));
    TRY(comm_ring(c, 1));
    COMM_HIP(c, hipMemcpyAsync(c->d_send, value, sizeof(long long), hipMemcpyHostToDevice, c->stream));
    TRY(comm_reduce_window(c, 0));
    COMM_HIP(c, hipEventSynchronize(c->ev[0]));
    *value = c->h_ring[0];
    return SCP_OK;
}

#define SCP_MAX_PARTS 16
struct CountPtrs { const int* p[SCP_MAX_PARTS]; int n; };
__global__ void sum_counts_kernel(CountPtrs a, long long* out)
{
    long long s = 0;
    for (int i = 0; i < a.n; i++) s += a.p[i] ? (long long)*a.p[i] : 0;
    *out = s;
}

extern "C" int scp_ptr_run_sharded(scp_comm_handle c, scp_handle* parts, int nparts, int lookahead, int* iterations, int* collectives)
{
    if (!parts || nparts < 1 || nparts > SCP_MAX_PARTS || lookahead < 1) return SCP_ERR_BAD_ARGUMENT;
    scp_comm* own = nullptr;      // comm == NULL: a private single-process communicator for the duration of the call
    if (!c) { TRY(scp_comm_create(nullptr, 0, 1, parts[0] ? parts[0]->device : 0, &own)); c = own; }
    struct Own { scp_comm* c; ~Own() { if (c) { g_comm_err = c->err; scp_comm_destroy(c); } } } own_guard{own};   // (errors of the private communicator stay readable: scp_comm_last_error(NULL))
    int iter_max = -1;
    for (int i = 0; i < nparts; i++) {
        scp_problem* h = parts[i];
        if (!h || !h->run_ready || h->B < 1 || h->device != c->device) { c->err = "scp_ptr_run_sharded: every part needs an initialised PTR run on the communicator's device"; return SCP_ERR_BAD_ARGUMENT; }
        if (iter_max >= 0 && h->pars.iter_max != iter_max) { c->err = "scp_ptr_run_sharded: parts with different iter_max"; return SCP_ERR_BAD_ARGUMENT; }
        iter_max = h->pars.iter_max;
    }
    COMM_HIP(c, hipSetDevice(c->device));
    const int windows = sharded_windows(iter_max, lookahead);
    TRY(comm_ring(c, windows));
    while ((int)c->part_ev.size() < nparts) { hipEvent_t e; COMM_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->part_ev.push_back(e); }
    const int it0 = parts[0]->iter;      // a run may be continued: windows count from the parts' current iteration
    int ncoll = 0;
    auto enqueue_window = [&](int w) -> int {
        for (int l = 0; l < lookahead; l++)
            for (int i = 0; i < nparts; i++) {
                const int rc = scp_ptr_iterate_async(parts[i]);
                if (rc) { c->err = std::string("scp_ptr_iterate_async: ") + parts[i]->err; return rc; }
            }
        CountPtrs cp{};
        cp.n = nparts;
        for (int i = 0; i < nparts; i++) {
            scp_problem* h = parts[i];
            // the count of the window's LAST iteration; nothing is enqueued beyond iter_max: the count there is 0
            cp.p[i] = (h->iter <= h->pars.iter_max && h->na_dev) ? h->na_dev + h->iter : nullptr;
            COMM_HIP(c, hipEventRecord(c->part_ev[i], h->stream));
            COMM_HIP(c, hipStreamWaitEvent(c->stream, c->part_ev[i], 0));
        }
        hipLaunchKernelGGL(sum_counts_kernel, dim3(1), dim3(1), 0, c->stream, cp, c->d_send + w);
        COMM_HIP(c, hipGetLastError());
        ncoll += c->comm ? 1 : 0;
        return comm_reduce_window(c, w);
    };
    // the window loop itself is csrc/sharded_loop.hpp: the code the world-size-2 gloo test drives on the CPU
    int done_window = -1;
    auto wait_window = [&](int w, long long* n) -> int {
        COMM_HIP(c, hipEventSynchronize(c->ev[w]));
        *n = c->h_ring[w];
        return SCP_OK;
    };
    // a rank that failed while enqueuing a window tells the others through that window's (and the next one's) all-reduce: sharded_loop.hpp
    auto abort_window = [&](int w) -> int {
        const long long sentinel = SHARDED_SENTINEL;
        if (hipMemcpyAsync(c->d_send + w, &sentinel, sizeof(long long), hipMemcpyHostToDevice, c->stream) != hipSuccess) return SCP_ERR_HIP;
        if (hipStreamSynchronize(c->stream) != hipSuccess) return SCP_ERR_HIP;      // (the source is on this stack frame)
        std::string keep = c->err;
        const int rc = comm_reduce_window(c, w);
        c->err = keep;          // the error to report is the one that made this rank fail
        return rc;
    };
    {
        const int rc = sharded_window_loop(windows, enqueue_window, wait_window, abort_window, &done_window);
        if (rc == SHARDED_PEER_FAILED) { c->err = "scp_ptr_run_sharded: another rank failed inside window " + std::to_string(done_window); return SCP_ERR_PEER; }
        if (rc) return rc;
    }
    for (int i = 0; i < nparts; i++) stamps_collect_ready(parts[i]);
    if (iterations) *iterations = sharded_iterations(it0, done_window, lookahead, iter_max);
    if (collectives) *collectives = ncoll;
    return SCP_OK;
}
