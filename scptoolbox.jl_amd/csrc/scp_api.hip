// C-ABI implementation (include/scp_mi355x.h) of the MI355X-native SCP inner loop.
// gfx950 only: no CUDA shims, no dual paths.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/scp_mi355x.h"
#include "discretize_kernel.hpp"
#include "models/double_integrator.hpp"
#include "models/quadrotor.hpp"
#include "models/rocket_landing.hpp"

using namespace scp;

struct scp_problem {
    int model_id = -1;
    scp_model_info info{};
    int N = 0, Nsub = 0, method = 0, cap = 0, device = 0;
    double feas_tol = 0;
    std::vector<double> par;
    std::vector<double> Sx, cx, Su, cu, Sp, cp;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // device scratch for the host-pointer entry points
    double *d_xd = nullptr, *d_ud = nullptr, *d_p = nullptr;
    double *d_A = nullptr, *d_Bm = nullptr, *d_Bp = nullptr, *d_F = nullptr, *d_r = nullptr, *d_E = nullptr;
    double *d_defect = nullptr, *d_iSx = nullptr;
    int* d_feas = nullptr;
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
    i->npar = M::npar;
}

extern "C" int scp_model_query(int model_id, scp_model_info* info)
{
    if (!info) return SCP_ERR_BAD_ARGUMENT;
    switch (model_id) {
        case SCP_MODEL_DOUBLE_INTEGRATOR: fill_info<DoubleIntegrator>(info); return SCP_OK;
        case SCP_MODEL_QUADROTOR: fill_info<Quadrotor>(info); return SCP_OK;
        case SCP_MODEL_ROCKET_LANDING: fill_info<RocketLanding>(info); return SCP_OK;
        default: return SCP_ERR_UNKNOWN_MODEL;
    }
}

extern "C" const char* scp_last_error(scp_handle h) { return h ? h->err.c_str() : "null handle"; }

static void free_all(scp_problem* h)
{
    double** bufs[] = {&h->d_xd, &h->d_ud, &h->d_p, &h->d_A, &h->d_Bm, &h->d_Bp, &h->d_F,
                       &h->d_r, &h->d_E, &h->d_defect, &h->d_iSx};
    for (auto b : bufs)
        if (*b) { (void)hipFree(*b); *b = nullptr; }
    if (h->d_feas) { (void)hipFree(h->d_feas); h->d_feas = nullptr; }
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
}

extern "C" int scp_problem_create(const scp_problem_desc* d, scp_handle* out)
{
    if (!d || !out) return SCP_ERR_BAD_ARGUMENT;
    *out = nullptr;
    scp_model_info info;
    int rc = scp_model_query(d->model_id, &info);
    if (rc) return rc;
    if (d->N < 2 || d->Nsub < 2 || d->batch_capacity < 1) return SCP_ERR_BAD_ARGUMENT;
    if (d->disc_method != SCP_FOH) return SCP_ERR_UNSUPPORTED;
    if (!d->model_par || !d->scale.Sx || !d->scale.cx || !d->scale.Su || !d->scale.cu) return SCP_ERR_BAD_ARGUMENT;
    if (info.np > 0 && (!d->scale.Sp || !d->scale.cp)) return SCP_ERR_BAD_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SCP_ERR_NO_DEVICE;
    if (d->device < 0 || d->device >= ndev) return SCP_ERR_NO_DEVICE;
    scp_problem* h = new (std::nothrow) scp_problem();
    if (!h) return SCP_ERR_ALLOC;
    h->model_id = d->model_id; h->info = info; h->N = d->N; h->Nsub = d->Nsub; h->method = d->disc_method;
    h->cap = d->batch_capacity; h->device = d->device; h->feas_tol = d->feas_tol;
    h->par.assign(d->model_par, d->model_par + info.npar);
    h->Sx.assign(d->scale.Sx, d->scale.Sx + info.nx); h->cx.assign(d->scale.cx, d->scale.cx + info.nx);
    h->Su.assign(d->scale.Su, d->scale.Su + info.nu); h->cu.assign(d->scale.cu, d->scale.cu + info.nu);
    if (info.np > 0) {
        h->Sp.assign(d->scale.Sp, d->scale.Sp + info.np); h->cp.assign(d->scale.cp, d->scale.cp + info.np);
    }
    *out = h;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_TRY(h, hipEventCreate(&h->ev0));
    HIP_TRY(h, hipEventCreate(&h->ev1));
    const size_t nx = info.nx, nu = info.nu, np = info.np > 0 ? info.np : 1, npF = info.npF > 0 ? info.npF : 1;
    const size_t B = h->cap, N = h->N, M = N - 1, D = sizeof(double);
    HIP_TRY(h, hipMalloc(&h->d_xd, nx * N * B * D));
    HIP_TRY(h, hipMalloc(&h->d_ud, nu * N * B * D));
    HIP_TRY(h, hipMalloc(&h->d_p, np * B * D));
    HIP_TRY(h, hipMalloc(&h->d_A, nx * nx * M * B * D));
    HIP_TRY(h, hipMalloc(&h->d_Bm, nx * nu * M * B * D));
    HIP_TRY(h, hipMalloc(&h->d_Bp, nx * nu * M * B * D));
    HIP_TRY(h, hipMalloc(&h->d_F, nx * npF * M * B * D));
    HIP_TRY(h, hipMalloc(&h->d_r, nx * M * B * D));
    HIP_TRY(h, hipMalloc(&h->d_E, nx * nx * M * B * D));
    HIP_TRY(h, hipMalloc(&h->d_defect, nx * M * B * D));
    HIP_TRY(h, hipMalloc(&h->d_feas, B * sizeof(int)));
    HIP_TRY(h, hipMalloc(&h->d_iSx, nx * D));
    std::vector<double> iSx(nx);
    for (size_t i = 0; i < nx; i++) iSx[i] = 1.0 / h->Sx[i];  // iSx = inv(Sx), scp.jl:492-493
    HIP_TRY(h, hipMemcpy(h->d_iSx, iSx.data(), nx * D, hipMemcpyHostToDevice));
    return SCP_OK;
}

extern "C" int scp_problem_destroy(scp_handle h)
{
    if (!h) return SCP_ERR_BAD_ARGUMENT;
    (void)hipSetDevice(h->device);
    free_all(h);
    delete h;
    return SCP_OK;
}

extern "C" int scp_sync(scp_handle h)
{
    if (!h) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return SCP_OK;
}

template <class M>
static int launch_discretize(scp_problem* h, const DiscArgs& a)
{
    using L = DiscLayout<M>;
    const long groups = (long)a.B * (a.N - 1);
    const int blocks = (int)((groups + L::GROUPS_PER_BLOCK - 1) / L::GROUPS_PER_BLOCK);
    typename M::Params P = M::make_params(h->par.data());
    hipLaunchKernelGGL(discretize_foh_kernel<M>, dim3(blocks), dim3(256), 0, h->stream, a, P);
    HIP_TRY(h, hipGetLastError());
    return SCP_OK;
}

static int discretize_dev(scp_problem* h, int B, const double* xd, const double* ud, const double* p, double* A,
                          double* Bm, double* Bp, double* F, double* r, double* E, double* defect, int* feas)
{
    DiscArgs a;
    a.B = B; a.N = h->N; a.Nsub = h->Nsub;
    a.xd = xd; a.ud = ud; a.p = p; a.iSx = h->d_iSx; a.feas_tol = h->feas_tol;
    a.A = A; a.Bm = Bm; a.Bp = Bp; a.F = F; a.r = r; a.E = E; a.defect = defect; a.feas = feas;
    HIP_TRY(h, hipMemsetAsync(feas, 0xff, (size_t)B * sizeof(int), h->stream));  // ref.feas = true (:179); any non-zero == true
    switch (h->model_id) {
        case SCP_MODEL_DOUBLE_INTEGRATOR: return launch_discretize<DoubleIntegrator>(h, a);
        case SCP_MODEL_QUADROTOR: return launch_discretize<Quadrotor>(h, a);
        case SCP_MODEL_ROCKET_LANDING: return launch_discretize<RocketLanding>(h, a);
        default: return SCP_ERR_UNKNOWN_MODEL;
    }
}

extern "C" int scp_discretize_batch_dev(scp_handle h, int B, const double* xd, const double* ud, const double* p,
                                        double* A, double* Bm, double* Bp, double* F, double* r, double* E,
                                        double* defect, int32_t* feas)
{
    if (!h || B < 1 || !xd || !ud || !A || !Bm || !Bp || !F || !r || !E || !defect || !feas) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    return discretize_dev(h, B, xd, ud, p, A, Bm, Bp, F, r, E, defect, feas);
}

extern "C" int scp_discretize_batch_host(scp_handle h, int B, const double* xd, const double* ud, const double* p,
                                         double* A, double* Bm, double* Bp, double* F, double* r, double* E,
                                         double* defect, uint8_t* feas, double* seconds)
{
    if (!h || B < 1 || !xd || !ud) return SCP_ERR_BAD_ARGUMENT;
    if (B > h->cap) return SCP_ERR_BATCH_TOO_LARGE;
    if (h->info.np > 0 && !p) return SCP_ERR_BAD_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nx = h->info.nx, nu = h->info.nu, np = h->info.np, npF = h->info.npF;
    const size_t N = h->N, M = N - 1, D = sizeof(double), b = B;
    HIP_TRY(h, hipMemcpyAsync(h->d_xd, xd, nx * N * b * D, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_ud, ud, nu * N * b * D, hipMemcpyHostToDevice, h->stream));
    if (np > 0) HIP_TRY(h, hipMemcpyAsync(h->d_p, p, np * b * D, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
    int rc = discretize_dev(h, B, h->d_xd, h->d_ud, h->d_p, h->d_A, h->d_Bm, h->d_Bp, h->d_F, h->d_r, h->d_E,
                            h->d_defect, h->d_feas);
    if (rc) return rc;
    HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
    if (A) HIP_TRY(h, hipMemcpyAsync(A, h->d_A, nx * nx * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (Bm) HIP_TRY(h, hipMemcpyAsync(Bm, h->d_Bm, nx * nu * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (Bp) HIP_TRY(h, hipMemcpyAsync(Bp, h->d_Bp, nx * nu * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (F && npF > 0) HIP_TRY(h, hipMemcpyAsync(F, h->d_F, nx * npF * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (r) HIP_TRY(h, hipMemcpyAsync(r, h->d_r, nx * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (E) HIP_TRY(h, hipMemcpyAsync(E, h->d_E, nx * nx * M * b * D, hipMemcpyDeviceToHost, h->stream));
    if (defect) HIP_TRY(h, hipMemcpyAsync(defect, h->d_defect, nx * M * b * D, hipMemcpyDeviceToHost, h->stream));
    std::vector<int> hf(B);
    HIP_TRY(h, hipMemcpyAsync(hf.data(), h->d_feas, b * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (feas)
        for (int i = 0; i < B; i++) feas[i] = hf[i] != 0;
    if (seconds) {
        float ms = 0;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        *seconds = ms * 1e-3;
    }
    return SCP_OK;
}
