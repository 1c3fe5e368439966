// Stage-form data layout of the reduced PTR subproblem + K2, the assembly kernel.
//
// The reference builds a fresh JuMP model every iteration (src/solvers/ptr.jl:213-293 with
// scp.jl:657-895 and ptr.jl:565-895).  Here the subproblem has a FIXED stage structure per
// (model, N); K2 only refills its values from the reference trajectory, the discrete dynamics
// produced by K1 and the model's Jacobians -- one thread per (problem, node).
//
// Reduced form (equivalent to the reference's conic program, see DESIGN.md "Reduction"):
//   variables  z_k = (xh_k, uh_k) (scaled, x = Sx xh + cx), p-hat
//   min  1/2 z'Qd z + q'z + (p part) + sum_k g_k(rows_k)
//   dyn_k : a = D_k z_k + E_k z_{k+1} + Fp_k p + cd_k      cost  om_k' |a|        (ptr.jl:813-887 P_k, vd eliminated)
//   tr_k  : a = z_k - zref_k                              cost  ttr_k (||a_x||_inf + ||a_u||_inf)   (ptr.jl:565-743)
//   trp   : a = p - pref                                  cost  ttrp ||a||_inf
//   loc_k : a = Kl_k z_k + Kp_k p + cl_k ; rows [0,ns) hinge cost hw_k' max(a,0) (scp.jl:744-794, vs eliminated),
//           rows [ns,ns+nl) a <= 0, then nsoc cones a in Q^4                      (scp.jl:685-734)
//   glin  : Lp p + lp <= 0
//   ic/tc : a = H x_1|x_N + K p + l      cost  bw' |a|                            (scp.jl:808-895, vic/vtc eliminated)
// All matrices are ROW-MAJOR [row][col]; one contiguous slab of doubles per problem.
#pragma once
#include <hip/hip_runtime.h>

namespace scp {

template <class M>
struct SP {
    static constexpr int nx = M::nx, nu = M::nu, np = M::np, nz = nx + nu;
    static constexpr int npa = np > 0 ? np : 1;
    static constexpr int ns = M::ns, nl = M::nl, nsoc = M::nsoc, ml = ns + nl + 4 * nsoc;
    static constexpr int ng = M::ng, nic = M::nic, ntc = M::ntc, nbc = nic > ntc ? nic : ntc;
    static constexpr int npp = M::npp;
    // per-stage row record (slacks / duals / residuals)
    static constexpr int R_DYN0 = 0, R_DYN1 = nx, R_H0 = 2 * nx, R_H1 = 2 * nx + ns, R_TR0 = 2 * nx + 2 * ns,
                         R_TR1 = R_TR0 + nz, R_LIN = R_TR1 + nz, R_SOC = R_LIN + nl, RS = R_SOC + 4 * nsoc;
    // global row record
    static constexpr int G_IC0 = 0, G_IC1 = nic, G_TC0 = 2 * nic, G_TC1 = 2 * nic + ntc, G_TRP0 = 2 * nic + 2 * ntc,
                         G_TRP1 = G_TRP0 + np, G_LIN = G_TRP1 + np, RG = G_LIN + ng;
    // per-stage aux record: [y_dyn (nx) | v (ns) | eta_x | eta_u]
    static constexpr int A_Y = 0, A_V = nx, A_EX = nx + ns, A_EU = nx + ns + 1, AS = nx + ns + 2;
    // global aux record: [y_ic | y_tc | eta_p]
    static constexpr int GA_YIC = 0, GA_YTC = nic, GA_EP = nic + ntc, AG = nic + ntc + 1;
    // augmented (nu) block rows of a stage: [dyn (nx) | hinge (ns) | bc (nbc: ic at k=0, tc at k=N-1)]
    static constexpr int MNU_MID = nx + ns, MNU = nx + ns + nbc;

    // ---- problem-data slab: N contiguous STAGE RECORDS followed by one GLOBAL RECORD (doubles) ----
    // stage record k (everything a sweep needs for node k, fetched with a few coalesced loads):
    static constexpr int O_QD = 0, O_Q = nz, O_ZREF = 2 * nz, O_TTR = 3 * nz, O_CD = O_TTR + 1, O_OM = O_CD + nx,
                         O_HW = O_OM + nx, O_CL = O_HW + (ns > 0 ? ns : 1), O_D = O_CL + ml, O_E = O_D + nx * nz,
                         O_FP = O_E + nx * nz, O_KL = O_FP + nx * npa, O_KP = O_KL + ml * nz,
                         SR = (O_KP + ml * npa + 1) & ~1;
    // global record
    static constexpr int Q_QP = 0, Q_QPL = npa, Q_PREF = 2 * npa, Q_LP = 3 * npa, Q_LPC = Q_LP + (ng > 0 ? ng : 1) * npa,
                         Q_H0 = Q_LPC + (ng > 0 ? ng : 1), Q_K0 = Q_H0 + nic * nx, Q_L0 = Q_K0 + nic * npa,
                         Q_BW0 = Q_L0 + nic, Q_HF = Q_BW0 + nic, Q_KF = Q_HF + ntc * nx, Q_LF = Q_KF + ntc * npa,
                         Q_BWF = Q_LF + ntc, Q_SCAL = Q_BWF + ntc, GR = (Q_SCAL + 2 + 1) & ~1;
    struct Off {
        long glob, total;
        // element accessors (kept in the style base + offset so that kernels read P[o.X(k) + i])
        __host__ __device__ long st(int k) const { return (long)k * SR; }
        __host__ __device__ long Qd(int k) const { return st(k) + O_QD; }
        __host__ __device__ long q(int k) const { return st(k) + O_Q; }
        __host__ __device__ long zref(int k) const { return st(k) + O_ZREF; }
        __host__ __device__ long ttr(int k) const { return st(k) + O_TTR; }
        __host__ __device__ long cd(int k) const { return st(k) + O_CD; }
        __host__ __device__ long om(int k) const { return st(k) + O_OM; }
        __host__ __device__ long hw(int k) const { return st(k) + O_HW; }
        __host__ __device__ long cl(int k) const { return st(k) + O_CL; }
        __host__ __device__ long D(int k) const { return st(k) + O_D; }
        __host__ __device__ long E(int k) const { return st(k) + O_E; }
        __host__ __device__ long Fp(int k) const { return st(k) + O_FP; }
        __host__ __device__ long Kl(int k) const { return st(k) + O_KL; }
        __host__ __device__ long Kp(int k) const { return st(k) + O_KP; }
        long Qp, qp, pref, Lp, lp, H0, K0, l0, bw0, Hf, Kf, lf, bwf, scal;
    };
    __host__ __device__ static Off offsets(int N)
    {
        Off o;
        o.glob = (long)N * SR;
        o.Qp = o.glob + Q_QP; o.qp = o.glob + Q_QPL; o.pref = o.glob + Q_PREF; o.Lp = o.glob + Q_LP; o.lp = o.glob + Q_LPC;
        o.H0 = o.glob + Q_H0; o.K0 = o.glob + Q_K0; o.l0 = o.glob + Q_L0; o.bw0 = o.glob + Q_BW0;
        o.Hf = o.glob + Q_HF; o.Kf = o.glob + Q_KF; o.lf = o.glob + Q_LF; o.bwf = o.glob + Q_BWF; o.scal = o.glob + Q_SCAL;
        o.total = (o.glob + GR + 7) & ~7L;
        return o;
    }
};

struct AsmArgs {
    int B, N;
    double wvc, wtr;
    const double* xd;  // [B][N][nx]  reference trajectory
    const double* ud;  // [B][N][nu]
    const double* p;   // [B][np]
    const double* pp;  // [B][npp] per-problem data
    const double* A;   // K1 outputs, column-major blocks
    const double* Bm;
    const double* Bp;
    const double* F;
    const double* r;
    const double *Sx, *cx, *Su, *cu, *Sp, *cp;  // device copies of the scaling
    double* slab;                               // [B][total]
    long slab_stride;
    const int* active;                          // optional [B]
};

__host__ __device__ __forceinline__ double linrange01(int N, int j)
{
    const double tt = (double)j / (double)(N - 1);
    return (1.0 - tt) * 0.0 + tt * 1.0;
}

// trapz weights on the uniform grid (src/utils/helper.jl:560-568): w_k = sum of adjacent half-intervals
__host__ __device__ __forceinline__ double trapz_w(int N, int k)
{
    double w = 0.0;
    if (k > 0) w += 0.5 * (linrange01(N, k) - linrange01(N, k - 1));
    if (k < N - 1) w += 0.5 * (linrange01(N, k + 1) - linrange01(N, k));
    return w;
}

// Fills stage record k (k < N) or the global record (k == N) of problem b.  __host__ __device__: the device kernel
// below calls it per thread; the CPU baseline (oracle/cpu_ptr.cpp, bench.py's cpu_baseline leg) calls the same code on
// the host so that both solvers are timed on identical subproblem data.
template <class M>
__host__ __device__ inline void ptr_assemble_entry(const AsmArgs& a, const typename M::Params& par, int b, int k)
{
    using S = SP<M>;
    constexpr int nx = S::nx, nu = S::nu, np = S::np, npa = S::npa, nz = S::nz, ns = S::ns, nl = S::nl, nsoc = S::nsoc,
                  ng = S::ng, nic = S::nic, ntc = S::ntc;
    const int N = a.N;
    const typename S::Off o = S::offsets(N);
    double* P = a.slab + (long)b * a.slab_stride;
    const double* xr = a.xd + (long)b * N * nx;
    const double* ur = a.ud + (long)b * N * nu;
    const double* pr = a.p + (long)b * np;
    const double* pp = a.pp + (long)b * S::npp;
    double Sz[nz], cz[nz];
    for (int i = 0; i < nx; i++) { Sz[i] = a.Sx[i]; cz[i] = a.cx[i]; }
    for (int i = 0; i < nu; i++) { Sz[nx + i] = a.Su[i]; cz[nx + i] = a.cu[i]; }
    double Spv[npa], cpv[npa];
    for (int i = 0; i < npa; i++) { Spv[i] = np > 0 ? a.Sp[i] : 1.0; cpv[i] = np > 0 ? a.cp[i] : 0.0; }

    double Qu[nu], lu[nu], lx[nx], tx[nx], tp[npa], Qp[npa];
    for (int i = 0; i < npa; i++) { tp[i] = 0.0; Qp[i] = 0.0; }
    M::cost_terms(par, Qu, lu, lx, tx, tp, Qp);

    if (k == N) {
        // ---- p cost, trust region reference, global rows, boundary conditions, constants ----
        double cc = 0.0;
        for (int i = 0; i < npa; i++) {
            P[o.Qp + i] = np > 0 ? 2.0 * Qp[i] * Spv[i] * Spv[i] : 0.0;
            P[o.qp + i] = np > 0 ? tp[i] * Spv[i] + 2.0 * Qp[i] * cpv[i] * Spv[i] : 0.0;
            P[o.pref + i] = np > 0 ? (pr[i] - cpv[i]) / Spv[i] : 0.0;
            if (np > 0) cc += tp[i] * cpv[i] + Qp[i] * cpv[i] * cpv[i];
        }
        for (int kk = 0; kk < N; kk++) {
            const double w = trapz_w(N, kk);
            for (int i = 0; i < nu; i++) cc += w * (Qu[i] * a.cu[i] * a.cu[i] + lu[i] * a.cu[i]);
            for (int i = 0; i < nx; i++) cc += w * lx[i] * a.cx[i];
        }
        for (int i = 0; i < nx; i++) cc += tx[i] * a.cx[i];
        P[o.scal + 0] = a.wtr;  // ttrp
        P[o.scal + 1] = cc;     // cost constant
        if (ng > 0) {
            double Lp[(ng > 0 ? ng : 1) * npa], lp[ng > 0 ? ng : 1];
            M::glin_rows(par, Lp, lp);
            for (int i = 0; i < ng; i++) {
                double n2 = 0.0, c0 = lp[i];
                for (int j = 0; j < np; j++) { const double v = Lp[i * npa + j] * Spv[j]; n2 += v * v; c0 += Lp[i * npa + j] * cpv[j]; }
                const double e = 1.0 / fmax(sqrt(n2), 1e-12);
                for (int j = 0; j < np; j++) P[o.Lp + i * npa + j] = Lp[i * npa + j] * Spv[j] * e;
                P[o.lp + i] = c0 * e;
            }
        }
        // boundary conditions (scp.jl:808-895): l = g - H xb - K pb ; rows scaled to unit norm
        for (int which = 0; which < 2; which++) {
            const int nb = which == 0 ? nic : ntc;
            const double* xb = which == 0 ? xr : xr + (long)(N - 1) * nx;
            double g[S::nbc], H[S::nbc * nx], K[S::nbc * npa];
            for (int i = 0; i < S::nbc * npa; i++) K[i] = 0.0;
            if (which == 0) M::bc_ic(par, xb, pr, pp, g, H, K);
            else M::bc_tc(par, xb, pr, pp, g, H, K);
            const long oH = which == 0 ? o.H0 : o.Hf, oK = which == 0 ? o.K0 : o.Kf, ol = which == 0 ? o.l0 : o.lf,
                       ow = which == 0 ? o.bw0 : o.bwf;
            for (int i = 0; i < nb; i++) {
                double l = g[i], n2 = 0.0;
                for (int j = 0; j < nx; j++) l -= H[i * nx + j] * xb[j];
                for (int j = 0; j < np; j++) l -= K[i * npa + j] * pr[j];
                double c0 = l;
                for (int j = 0; j < nx; j++) { const double v = H[i * nx + j] * a.Sx[j]; n2 += v * v; c0 += H[i * nx + j] * a.cx[j]; }
                for (int j = 0; j < np; j++) { const double v = K[i * npa + j] * Spv[j]; n2 += v * v; c0 += K[i * npa + j] * cpv[j]; }
                const double e = 1.0 / fmax(sqrt(n2), 1e-12);
                for (int j = 0; j < nx; j++) P[oH + i * nx + j] = H[i * nx + j] * a.Sx[j] * e;
                for (int j = 0; j < npa; j++) P[oK + i * npa + j] = np > 0 ? K[i * npa + j] * Spv[j] * e : 0.0;
                P[ol + i] = c0 * e;
                P[ow + i] = a.wvc / e;
            }
        }
        return;
    }

    const double w = trapz_w(N, k);
    const double tk = linrange01(N, k);
    // ---- cost (scp.jl:552-601) ----
    for (int i = 0; i < nx; i++) {
        P[o.Qd(k) + i] = 0.0;
        P[o.q(k) + i] = w * lx[i] * a.Sx[i] + (k == N - 1 ? tx[i] * a.Sx[i] : 0.0);
    }
    for (int i = 0; i < nu; i++) {
        P[o.Qd(k) + nx + i] = 2.0 * w * Qu[i] * a.Su[i] * a.Su[i];
        P[o.q(k) + nx + i] = w * (2.0 * Qu[i] * a.cu[i] * a.Su[i] + lu[i] * a.Su[i]);
    }
    // ---- trust region reference ----
    for (int i = 0; i < nx; i++) P[o.zref(k) + i] = (xr[(long)k * nx + i] - a.cx[i]) / a.Sx[i];
    for (int i = 0; i < nu; i++) P[o.zref(k) + nx + i] = (ur[(long)k * nu + i] - a.cu[i]) / a.Su[i];
    P[o.ttr(k)] = a.wtr * w;
    // ---- dynamics rows scaled by iSx (discretization.jl:458-467) ----
    if (k == N - 1) {
        for (int i = 0; i < nx * nz; i++) { P[o.D(k) + i] = 0.0; P[o.E(k) + i] = 0.0; }
        for (int i = 0; i < nx * npa; i++) P[o.Fp(k) + i] = 0.0;
        for (int i = 0; i < nx; i++) { P[o.cd(k) + i] = 0.0; P[o.om(k) + i] = 0.0; }
    }
    if (k < N - 1) {
        const long ik = (long)b * (N - 1) + k;
        const double* Ak = a.A + ik * nx * nx;    // column-major
        const double* Bmk = a.Bm + ik * nx * nu;
        const double* Bpk = a.Bp + ik * nx * nu;
        const double* Fk = a.F + ik * nx * (M::npF > 0 ? M::npF : 1);
        const double* rk = a.r + ik * nx;
        double* D = P + o.D(k);
        double* E = P + o.E(k);
        double* Fp = P + o.Fp(k);
        for (int i = 0; i < nx; i++) {
            const double is = 1.0 / a.Sx[i];
            double cphys = a.cx[i] - rk[i];
            for (int j = 0; j < nx; j++) {
                D[i * nz + j] = -(is * Ak[i + nx * j] * a.Sx[j]);
                E[i * nz + j] = (i == j) ? 1.0 : 0.0;
                cphys -= Ak[i + nx * j] * a.cx[j];
            }
            for (int j = 0; j < nu; j++) {
                D[i * nz + nx + j] = -(is * Bmk[i + nx * j] * a.Su[j]);
                E[i * nz + nx + j] = -(is * Bpk[i + nx * j] * a.Su[j]);
                cphys -= (Bmk[i + nx * j] + Bpk[i + nx * j]) * a.cu[j];
            }
            for (int j = 0; j < npa; j++) Fp[i * npa + j] = 0.0;
            for (int jj = 0; jj < M::npF; jj++) {
                const int j = M::Fcol(jj);
                Fp[i * npa + j] = -(is * Fk[i + nx * jj] * Spv[j]);
                cphys -= Fk[i + nx * jj] * cpv[j];
            }
            P[o.cd(k) + i] = is * cphys;
            P[o.om(k) + i] = a.wvc * w * a.Sx[i];
        }
    }
    // ---- stage-local rows ----
    double* Kl = P + o.Kl(k);
    double* Kp = P + o.Kp(k);
    double* cl = P + o.cl(k);
    const double* xk = xr + (long)k * nx;
    const double* uk = ur + (long)k * nu;
    if (ns > 0) {
        double s[ns > 0 ? ns : 1], C[(ns > 0 ? ns : 1) * nx], Dm[(ns > 0 ? ns : 1) * nu], G[(ns > 0 ? ns : 1) * npa];
        for (int i = 0; i < (ns > 0 ? ns : 1) * npa; i++) G[i] = 0.0;
        M::s_eval(par, tk, k + 1, xk, uk, pr, s, C, Dm, G);
        for (int i = 0; i < ns; i++) {
            double rr = s[i], n2 = 0.0;
            for (int j = 0; j < nx; j++) rr -= C[i * nx + j] * xk[j];
            for (int j = 0; j < nu; j++) rr -= Dm[i * nu + j] * uk[j];
            for (int j = 0; j < np; j++) rr -= G[i * npa + j] * pr[j];
            double c0 = rr;
            for (int j = 0; j < nx; j++) { const double v = C[i * nx + j] * a.Sx[j]; n2 += v * v; c0 += C[i * nx + j] * a.cx[j]; }
            for (int j = 0; j < nu; j++) { const double v = Dm[i * nu + j] * a.Su[j]; n2 += v * v; c0 += Dm[i * nu + j] * a.cu[j]; }
            for (int j = 0; j < np; j++) { const double v = G[i * npa + j] * Spv[j]; n2 += v * v; c0 += G[i * npa + j] * cpv[j]; }
            const double e = 1.0 / fmax(sqrt(n2), 1e-12);
            for (int j = 0; j < nx; j++) Kl[i * nz + j] = C[i * nx + j] * a.Sx[j] * e;
            for (int j = 0; j < nu; j++) Kl[i * nz + nx + j] = Dm[i * nu + j] * a.Su[j] * e;
            for (int j = 0; j < npa; j++) Kp[i * npa + j] = np > 0 ? G[i * npa + j] * Spv[j] * e : 0.0;
            cl[i] = c0 * e;
            P[o.hw(k) + i] = a.wvc * w / e;
        }
    }
    if (nl > 0) {
        double L[(nl > 0 ? nl : 1) * nz], Lp[(nl > 0 ? nl : 1) * npa], l[nl > 0 ? nl : 1];
        for (int i = 0; i < (nl > 0 ? nl : 1) * npa; i++) Lp[i] = 0.0;
        M::lin_rows(par, tk, k + 1, L, Lp, l);
        for (int i = 0; i < nl; i++) {
            double n2 = 0.0, c0 = l[i];
            for (int j = 0; j < nz; j++) { const double v = L[i * nz + j] * Sz[j]; n2 += v * v; c0 += L[i * nz + j] * cz[j]; }
            for (int j = 0; j < np; j++) { const double v = Lp[i * npa + j] * Spv[j]; n2 += v * v; c0 += Lp[i * npa + j] * cpv[j]; }
            const double e = n2 > 0.0 ? 1.0 / fmax(sqrt(n2), 1e-12) : 1.0;   // an all-zero (placeholder) row is left unscaled
            for (int j = 0; j < nz; j++) Kl[(ns + i) * nz + j] = L[i * nz + j] * Sz[j] * e;
            for (int j = 0; j < npa; j++) Kp[(ns + i) * npa + j] = np > 0 ? Lp[i * npa + j] * Spv[j] * e : 0.0;
            cl[ns + i] = c0 * e;
        }
    }
    if (nsoc > 0) {
        double Mm[(nsoc > 0 ? nsoc : 1) * 4 * nz], m[(nsoc > 0 ? nsoc : 1) * 4];
        M::soc_rows(par, tk, k + 1, Mm, m);
        for (int c = 0; c < nsoc; c++) {
            double mx = 0.0;
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < nz; j++) mx = fmax(mx, fabs(Mm[(4 * c + i) * nz + j] * Sz[j]));
            const double e = 1.0 / fmax(mx, 1e-12);  // one factor per cone keeps the cone a cone
            for (int i = 0; i < 4; i++) {
                const int row = ns + nl + 4 * c + i;
                double c0 = m[4 * c + i];
                for (int j = 0; j < nz; j++) { Kl[row * nz + j] = Mm[(4 * c + i) * nz + j] * Sz[j] * e; c0 += Mm[(4 * c + i) * nz + j] * cz[j]; }
                for (int j = 0; j < npa; j++) Kp[row * npa + j] = 0.0;
                cl[row] = c0 * e;
            }
        }
    }
}

template <class M>
__global__ __launch_bounds__(64) void ptr_assemble_kernel(AsmArgs a, typename M::Params par)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.B * (a.N + 1);
    if (tid >= total) return;
    const int b = (int)(tid / (a.N + 1));
    const int k = (int)(tid % (a.N + 1));  // k == N: global rows / boundary conditions / scalars
    if (a.active != nullptr && a.active[b] == 0) return;
    ptr_assemble_entry<M>(a, par, b, k);
}

}  // namespace scp
