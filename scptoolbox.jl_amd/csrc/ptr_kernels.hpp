// K4: per-problem reductions and PTR outer-loop logic on the device.
//   ptr_extract_kernel : un-scale the subproblem solution (value(blk), src/parser/block.jl:368-394), cost
//                        split J / J_tr / J_vc / J_aug (ptr.jl:753-895) and deviation (scp.jl:909-931)
//   ptr_update_kernel  : unsafe_solution (scp.jl:965-980), check_stopping_criterion! (ptr.jl:908-932),
//                        ref <- sol bookkeeping (ptr.jl:509), history record
#pragma once
#include <hip/hip_runtime.h>

#include "ipm_kernel.hpp"

namespace scp {

// per-iteration history record
enum { H_J = 0, H_JTR, H_JVC, H_JAUG, H_DEV, H_IMPROV, H_FEAS, H_STATUS, H_IPMIT, H_ACTIVE, H_GAP, H_PRES, H_DRES, H_N = 16 };

struct ExtractArgs {
    int B, N;
    const double* slab;
    long slab_stride;
    const double* z;   // [B][N*nz]
    const double* ph;  // [B][npa]
    const double *Sx, *cx, *Su, *cu, *Sp, *cp;
    const int* active;  // [B]
    double* xd;         // [B][N][nx]
    double* ud;         // [B][N][nu]
    double* p;          // [B][np]
    double* cost;       // [B][4]: J, J_tr, J_vc, J_aug
    double* dev;        // [B]
    double* eta;        // [B][2N+1]: eta_x[N], eta_u[N], eta_p  (sol.ηx, sol.ηu, sol.ηp)
    // virtual controls and penalty epigraphs the reference's SubproblemSolution stores (ptr.jl:399-432): they are
    // eliminated analytically from the reduced subproblem and re-evaluated from its solution here
    const double* Eref; // [B][N-1][nx*nx] column-major discretised E of the reference (ref.dyn.E, ptr.jl:805)
    double* vd;         // [B][N-1][nx]   E_k vd_k = linearised dynamics defect
    double* vs;         // [B][N][ns]     max(linearised s, 0)
    double* vic;        // [B][nic]
    double* vtc;        // [B][ntc]
    double* Ppen;       // [B][N]         P_k = ||E_k vd_k||_1 + ||vs_k||_1  (ptr.jl:813-887)
    double* Pf;         // [B][2]         ||vic||_1, ||vtc||_1
    double wvc;
};

// x = M^-1 b for a small column-major n x n matrix (Gaussian elimination with partial pivoting, one thread)
template <int n>
__device__ __forceinline__ void small_solve(const double* Mcm, const double* b, double* x)
{
    double A[n][n + 1];
#pragma unroll
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < n; j++) A[i][j] = Mcm[i + n * j];
        A[i][n] = b[i];
    }
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int i = c + 1; i < n; i++) if (fabs(A[i][c]) > fabs(A[piv][c])) piv = i;
        if (piv != c) for (int j = c; j <= n; j++) { const double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        const double d = A[c][c] != 0.0 ? 1.0 / A[c][c] : 0.0;
        for (int i = c + 1; i < n; i++) {
            const double f = A[i][c] * d;
            for (int j = c; j <= n; j++) A[i][j] -= f * A[c][j];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        double acc = A[i][n];
        for (int j = i + 1; j < n; j++) acc -= A[i][j] * x[j];
        x[i] = A[i][i] != 0.0 ? acc / A[i][i] : 0.0;
    }
}

// the body of K4a for problem b by one wavefront -- the stand-alone kernel below, or the tail of the wave that solved the
// problem (ipm2_solve_kernel with IpmArgs::ext: the wave that holds the slot does the extraction instead of 2 048 new waves
// queueing for slots behind the other stream's solves -- 21 ms per launch under two concurrent streams against 0.27 ms alone)
// (CALLER: one instantiation per calling kernel, so that the kernel's occupancy attribute propagates to this function --
// a callee shared by kernels with different register budgets gets none, and the solver kernel would drop to one wave per SIMD)
template <class M, int CALLER>
__device__ __noinline__ void ptr_extract_body(const ExtractArgs& a, const int b, const int lane)
{
    using S = SP<M>;
    constexpr int nx = S::nx, nu = S::nu, np = S::np, npa = S::npa, nz = S::nz, ns = S::ns, nic = S::nic,
                  ntc = S::ntc;
    const int N = a.N;
    const typename S::Off o = S::offsets(N);
    const double* P = a.slab + (long)b * a.slab_stride;
    const double* z = a.z + (long)b * N * nz;
    const double* ph = a.ph + (long)b * npa;
    double J = 0.0, Jtr = 0.0, Jvc = 0.0, devx = 0.0;
    for (int k = lane; k < N; k += 64) {
        const double* zk = z + (long)k * nz;
        for (int i = 0; i < nx; i++) a.xd[((long)b * N + k) * nx + i] = a.Sx[i] * zk[i] + a.cx[i];
        for (int i = 0; i < nu; i++) a.ud[((long)b * N + k) * nu + i] = a.Su[i] * zk[nx + i] + a.cu[i];
        double ex = 0.0, eu = 0.0;
        for (int j = 0; j < nz; j++) {
            J += 0.5 * P[o.Qd(k) + j] * zk[j] * zk[j] + P[o.q(k) + j] * zk[j];
            const double d = fabs(zk[j] - P[o.zref(k) + j]);
            if (j < nx) ex = fmax(ex, d); else eu = fmax(eu, d);
        }
        Jtr += P[o.ttr(k)] * (ex + eu);
        devx = fmax(devx, ex);
        a.eta[(long)b * (2 * N + 1) + k] = ex;
        a.eta[(long)b * (2 * N + 1) + N + k] = eu;
        double Pk = 0.0;
        if (k < N - 1) {
            const double* zn = zk + nz;
            double Evd[nx], vdk[nx];
            for (int i = 0; i < nx; i++) {
                double acc = P[o.cd(k) + i];
                const double *d = P + o.D(k) + i * nz, *e = P + o.E(k) + i * nz;
                for (int j = 0; j < nz; j++) acc += d[j] * zk[j] + e[j] * zn[j];
                for (int j = 0; j < np; j++) acc += P[o.Fp(k) + i * npa + j] * ph[j];
                Jvc += P[o.om(k) + i] * fabs(acc);
                Evd[i] = a.Sx[i] * acc;   // rows are scaled by iSx (discretization.jl:458-467): E_k vd_k in physical units
                Pk += fabs(Evd[i]);
            }
            small_solve<nx>(a.Eref + ((long)b * (N - 1) + k) * nx * nx, Evd, vdk);
            for (int i = 0; i < nx; i++) a.vd[((long)b * (N - 1) + k) * nx + i] = vdk[i];
        }
        const double wk = trapz_w(N, k);
        for (int i = 0; i < ns; i++) {
            double acc = P[o.cl(k) + i];
            for (int j = 0; j < nz; j++) acc += P[o.Kl(k) + i * nz + j] * zk[j];
            for (int j = 0; j < np; j++) acc += P[o.Kp(k) + i * npa + j] * ph[j];
            Jvc += P[o.hw(k) + i] * fmax(acc, 0.0);
            const double vsi = fmax(acc, 0.0) * P[o.hw(k) + i] / (a.wvc * wk);   // undo the unit-norm row scaling
            a.vs[((long)b * N + k) * (ns > 0 ? ns : 1) + i] = vsi;
            Pk += vsi;
        }
        a.Ppen[(long)b * N + k] = Pk;
    }
    J = wave_sum(J); Jtr = wave_sum(Jtr); Jvc = wave_sum(Jvc); devx = wave_max(devx);
    if (lane == 0) {
        double ep = 0.0;
        for (int j = 0; j < np; j++) {
            a.p[(long)b * np + j] = a.Sp[j] * ph[j] + a.cp[j];
            J += 0.5 * P[o.Qp + j] * ph[j] * ph[j] + P[o.qp + j] * ph[j];
            ep = fmax(ep, fabs(ph[j] - P[o.pref + j]));
        }
        J += P[o.scal + 1];
        if (np > 0) Jtr += P[o.scal + 0] * ep;
        double pf0 = 0.0, pf1 = 0.0;
        for (int i = 0; i < nic; i++) {
            double acc = P[o.l0 + i];
            for (int j = 0; j < nx; j++) acc += P[o.H0 + i * nx + j] * z[j];
            for (int j = 0; j < np; j++) acc += P[o.K0 + i * npa + j] * ph[j];
            Jvc += P[o.bw0 + i] * fabs(acc);
            const double v = -acc * P[o.bw0 + i] / a.wvc;   // H0 x + K0 p + l0 + vic = 0 (scp.jl:823-851), row scaling undone
            a.vic[(long)b * nic + i] = v; pf0 += fabs(v);
        }
        for (int i = 0; i < ntc; i++) {
            double acc = P[o.lf + i];
            for (int j = 0; j < nx; j++) acc += P[o.Hf + i * nx + j] * z[(long)(N - 1) * nz + j];
            for (int j = 0; j < np; j++) acc += P[o.Kf + i * npa + j] * ph[j];
            Jvc += P[o.bwf + i] * fabs(acc);
            const double v = -acc * P[o.bwf + i] / a.wvc;
            a.vtc[(long)b * ntc + i] = v; pf1 += fabs(v);
        }
        a.Pf[(long)b * 2 + 0] = pf0; a.Pf[(long)b * 2 + 1] = pf1;
        a.cost[(long)b * 4 + 0] = J; a.cost[(long)b * 4 + 1] = Jtr; a.cost[(long)b * 4 + 2] = Jvc;
        a.cost[(long)b * 4 + 3] = J + Jtr + Jvc;
        a.dev[b] = ep + devx;  // ||dp||_inf + max_k ||dx_k||_inf   (q_exit = Inf)
        a.eta[(long)b * (2 * N + 1) + 2 * N] = ep;
    }
}

template <class M>
__global__ __launch_bounds__(64) void ptr_extract_kernel(ExtractArgs a)
{
    if (!a.active[blockIdx.x]) return;
    ptr_extract_body<M, 0>(a, blockIdx.x, threadIdx.x);
}

struct UpdateArgs {
    int B, iter, iter_max;
    double eps_abs, eps_rel;
    const double* cost;     // [B][4]
    const double* dev;      // [B]
    const int* feas;        // [B] of the new solution (discretize!)
    const int* ipm_status;  // [B]
    const int* ipm_iters;   // [B]
    const double* ipm_info; // [B][8]
    double* Jaug_ref;       // [B] (NaN for the initial guess, ptr.jl:350)
    int* active;            // [B]
    int* scp_status;        // [B]: 0 running/solved, 1 failed (unsafe solution)
    int* iters_done;        // [B]
    double* hist;           // [iter_max][B][H_N]
    int* n_active;          // [1]
};

__global__ void ptr_update_kernel(UpdateArgs a)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    double* h = a.hist + ((long)(a.iter - 1) * a.B + b) * H_N;
    if (!a.active[b]) { h[H_ACTIVE] = 0.0; return; }
    const double J = a.cost[(long)b * 4 + 0], Jtr = a.cost[(long)b * 4 + 1], Jvc = a.cost[(long)b * 4 + 2],
                 Jaug = a.cost[(long)b * 4 + 3];
    const double Jref = a.Jaug_ref[b];
    const double improv = (Jref - Jaug) / fabs(Jref);  // NaN at the first iteration like the reference
    const bool unsafe = a.ipm_status[b] > 1;           // not OPTIMAL / ALMOST_OPTIMAL (scp.jl:975)
    const bool feas = a.feas[b] != 0;
    const bool stop = a.iter > 1 && (feas && (fabs(improv) <= a.eps_rel || a.dev[b] <= a.eps_abs));  // ptr.jl:924-927
    h[H_J] = J; h[H_JTR] = Jtr; h[H_JVC] = Jvc; h[H_JAUG] = Jaug; h[H_DEV] = a.dev[b]; h[H_IMPROV] = improv;
    h[H_FEAS] = feas ? 1.0 : 0.0; h[H_STATUS] = (double)a.ipm_status[b]; h[H_IPMIT] = (double)a.ipm_iters[b];
    h[H_ACTIVE] = 1.0; h[H_GAP] = a.ipm_info[(long)b * 8 + 2]; h[H_PRES] = a.ipm_info[(long)b * 8 + 3];
    h[H_DRES] = a.ipm_info[(long)b * 8 + 4];
    a.iters_done[b] = a.iter;
    if (unsafe) { a.scp_status[b] = 1; a.active[b] = 0; return; }  // emergency exit (ptr.jl:488-491)
    a.Jaug_ref[b] = Jaug;                                          // ref = spbm.sol (ptr.jl:509)
    if (stop || a.iter >= a.iter_max) { a.active[b] = 0; return; }
    atomicAdd(a.n_active, 1);
}

// traj.guess(N) for a Monte-Carlo batch on the device (generate_initial_guess, src/solvers/ptr.jl:548-555 ->
// problem.jl:686-700): one thread per (problem, node); the per-problem data pp (initial / terminal conditions)
// is all that crosses PCIe.
struct GuessArgs {
    int B, N;
    const double* pp;   // [npp,B]
    double* xd;         // [nx,N,B]
    double* ud;         // [nu,N,B]
    double* p;          // [np + np_node N,B]
    const int* only = nullptr;   // optional [B]: only the problems with only[b] != 0 are written
};
template <class M>
__global__ __launch_bounds__(256) void ptr_guess_kernel(GuessArgs a, typename M::Params par)
{
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)a.B * a.N) return;
    const int b = (int)(gid / a.N), k = (int)(gid % a.N);
    if (a.only != nullptr && a.only[b] == 0) return;
    double x[M::nx], u[M::nu], pv[M::np > 0 ? M::np : 1], pn[M::np_node > 0 ? M::np_node : 1];
    M::guess(par, a.pp + (long)b * M::npp, a.N, k, x, u, pv, pn);
#pragma unroll
    for (int i = 0; i < M::nx; i++) a.xd[((long)b * a.N + k) * M::nx + i] = x[i];
#pragma unroll
    for (int i = 0; i < M::nu; i++) a.ud[((long)b * a.N + k) * M::nu + i] = u[i];
    double* pb = a.p + (long)b * np_total<M>(a.N);
    if (k == 0) {
#pragma unroll
        for (int i = 0; i < M::np; i++) pb[i] = pv[i];
    }
#pragma unroll
    for (int i = 0; i < M::np_node; i++) pb[M::np + M::np_node * k + i] = pn[i];     // the node's own parameters
}

}  // namespace scp
