// The reference's Starship initial guess (test/examples/starship_flip/definition.jl:97-445) ON THE DEVICE, per instance of a
// Monte-Carlo batch:
//   phase 1  bang-bang gimbal flip at minimum three-engine thrust, RK4 on LinRange(0, 2 ts + 10, 5000) without aerodynamic torques
//            (:120-171), cut where the vertical speed reaches the switch speed, resampled on the first half of the SCP grid
//            -- `starship_flip_kernel`, one thread per instance (two integrations: the first finds the cut, the second samples);
//   phase 2  terminal descent as a convex program on a double integrator (:183-231, 404-421): thrust-vector inputs, SOC thrust
//            and tilt bounds, FOH-discretised with the reference's own RK4 recipe, one program per candidate duration
//            t2 = 10, 11, ..., 40 s -- all (instance, duration) programs are ONE batch of the conic engine, their values
//            written by `starship_descent_fill_kernel`;
//   then     the first feasible duration of every instance is taken and theta, T, omega, m of phase 2 are reconstructed from
//            the thrust vectors (:423-440) -- `starship_reconstruct_kernel`.
// Host-side numpy twin: scptoolbox.jl_amd/starship_guess.py (the golden fixtures are generated with it through the oracle's
// solver); this file is what scp_guess_batch_host runs for the Starship model.  Included by scp_api.hip.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "conic_engine.hpp"
#include "models/starship.hpp"

namespace scp {

// constants of the guess that are not model parameters: phase-switch attitude and vertical speed (parameters.jl:184-185)
#define SG_THETA_S (-10.0 * 0.017453292519943295)
#define SG_VS_Y (-10.0)
static constexpr int SG_NFLIP = 5000;      // definition.jl:141
static constexpr int SG_NCAND = 31;        // t2 = 10 ... 40 s (:404-421)

struct SgDev {
    int B, N, n1, N2, id_sw;          // instances, grid, |id1|, |id2|, index of the switch node (last of id1 = first of id2)
    const double* pp;                 // [B][5]: r0(2) v0(2) theta0
    double *xd, *ud, *p;              // outputs, ABI layout [B][N][nx], [B][N][nu], [B][np]
    double* xs;                       // [B][8] state at the switch node
    double* t1;                       // [B]
    int* ok1;                         // [B] phase 1 found the crossing
};

__device__ __forceinline__ double sg_linrange(double a, double b, int n, int j)
{
    const double t = (double)j / (double)(n - 1);
    return (1.0 - t) * a + t * b;
}
// dynamics(...; no_aero_torques = true) with unit time dilation (definition.jl:498-550)
__device__ __forceinline__ void sg_flip_f(const Starship::Params& K, const double (&x)[8], double T, double de, double (&f)[8])
{
    const double th = x[4], om = x[5], dd = x[7];
    const double ei0 = cos(th), ei1 = sin(th), ej0 = -sin(th), ej1 = cos(th);
    const double Tv0 = T * (-sin(de) * ei0 + cos(de) * ej0), Tv1 = T * (-sin(de) * ei1 + cos(de) * ej1);
    const double MT = -K.lcg * T * sin(de);
    const double nv = sqrt(x[2] * x[2] + x[3] * x[3]);
    const double D0 = -K.CD * nv * x[2], D1 = -K.CD * nv * x[3];
    f[0] = x[2]; f[1] = x[3];
    f[2] = (Tv0 + D0) / K.m + 0.0; f[3] = (Tv1 + D1) / K.m + (-K.g0);
    f[4] = om; f[5] = MT / K.J; f[6] = K.alpha_e * T; f[7] = (de - dd) / K.rate_delay;
}
__device__ __forceinline__ double sg_ctrl_delta(const Starship::Params& K, double t, double ts)
{
    return t <= ts ? K.delta_max : (t <= 2 * ts ? -K.delta_max : 0.0);
}
__device__ __forceinline__ void sg_rk4_step(const Starship::Params& K, double t0, double h, double ts, double (&x)[8])
{
    double k1[8], k2[8], k3[8], k4[8], y[8];
    sg_flip_f(K, x, K.T_min3, sg_ctrl_delta(K, t0, ts), k1);
    for (int i = 0; i < 8; i++) y[i] = x[i] + h / 2 * k1[i];
    sg_flip_f(K, y, K.T_min3, sg_ctrl_delta(K, t0 + h / 2, ts), k2);
    for (int i = 0; i < 8; i++) y[i] = x[i] + h / 2 * k2[i];
    sg_flip_f(K, y, K.T_min3, sg_ctrl_delta(K, t0 + h / 2, ts), k3);
    for (int i = 0; i < 8; i++) y[i] = x[i] + h * k3[i];
    sg_flip_f(K, y, K.T_min3, sg_ctrl_delta(K, t0 + h, ts), k4);
    for (int i = 0; i < 8; i++) x[i] = x[i] + h / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}

__global__ void starship_flip_kernel(SgDev a, Starship::Params K)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const double* pp = a.pp + (long)b * 5;
    const double th0 = pp[4];
    const double flip_ac = K.lcg / K.J * K.T_min3 * sin(K.delta_max);
    const double ts = sqrt((th0 - SG_THETA_S) / flip_ac);
    const double tf = 2 * ts + 10.0;
    double x[8] = {pp[0], pp[1], pp[2], pp[3], th0, 0.0, 0.0, K.delta_max};
    // ---- first integration: index of the first sample with v_y >= vs_y (:160-171) ----
    int k0 = (x[3] >= SG_VS_Y) ? 0 : -1;
    for (int j = 1; j < SG_NFLIP && k0 < 0; j++) {
        const double t0 = sg_linrange(0.0, tf, SG_NFLIP, j - 1), t1 = sg_linrange(0.0, tf, SG_NFLIP, j);
        sg_rk4_step(K, t0, t1 - t0, ts, x);
        if (x[3] >= SG_VS_Y) k0 = j;
    }
    a.ok1[b] = k0 >= 0 ? 1 : 0;
    if (k0 < 0) return;        // "no terminal velocity crossing": the caller falls back to the straight-line guess
    const double t1 = sg_linrange(0.0, tf, SG_NFLIP, k0);
    a.t1[b] = t1;
    // ---- second integration: np.interp of the truncated trajectory at tq_k = tau_k / tau_s * t1, k in id1 ----
    double xa[8] = {pp[0], pp[1], pp[2], pp[3], th0, 0.0, 0.0, K.delta_max}, xb[8];
    for (int i = 0; i < 8; i++) xb[i] = xa[i];
    int j = 0;                 // xa = X[j], xb = X[j + 1] (when j < k0)
    double ta = 0.0, tb = 0.0;
    if (k0 > 0) { tb = sg_linrange(0.0, tf, SG_NFLIP, 1); sg_rk4_step(K, 0.0, tb - 0.0, ts, xb); }
    double* xo = a.xd + (long)b * a.N * 8;
    double* uo = a.ud + (long)b * a.N * 3;
    for (int k = 0; k < a.n1; k++) {
        const double tau = sg_linrange(0.0, 1.0, a.N, k);
        const double tq = tau / K.tau_s * t1;
        double v[8];
        if (k0 == 0 || tq >= t1) {
            // np.interp clamps to the last sample; advance to the end first
            while (j + 1 < k0) {
                for (int i = 0; i < 8; i++) xa[i] = xb[i];
                ta = tb; j++;
                tb = sg_linrange(0.0, tf, SG_NFLIP, j + 1);
                sg_rk4_step(K, ta, tb - ta, ts, xb);
            }
            for (int i = 0; i < 8; i++) v[i] = k0 == 0 ? xa[i] : xb[i];
        } else {
            while (tq >= tb && j + 1 < k0) {
                for (int i = 0; i < 8; i++) xa[i] = xb[i];
                ta = tb; j++;
                tb = sg_linrange(0.0, tf, SG_NFLIP, j + 1);
                sg_rk4_step(K, ta, tb - ta, ts, xb);
            }
            for (int i = 0; i < 8; i++) { const double slope = (xb[i] - xa[i]) / (tb - ta); v[i] = slope * (tq - ta) + xa[i]; }
        }
        for (int i = 0; i < 8; i++) xo[(long)k * 8 + i] = v[i];
        uo[(long)k * 3 + 0] = K.T_min3; uo[(long)k * 3 + 1] = sg_ctrl_delta(K, tq, ts); uo[(long)k * 3 + 2] = 0.0;
        if (k == a.n1 - 1) for (int i = 0; i < 8; i++) a.xs[(long)b * 8 + i] = v[i];
    }
    for (int k = a.n1; k < a.N; k++) {
        for (int i = 0; i < 8; i++) xo[(long)k * 8 + i] = 0.0;
        for (int i = 0; i < 3; i++) uo[(long)k * 3 + i] = 0.0;
    }
}

// scaling of the descent program from the switch state (definition.jl:233-262: widths below sqrt(eps) are left at 1)
__host__ __device__ __forceinline__ void sg_scale(double lo, double hi, double& S, double& c)
{
    S = 1.0; c = 0.0;
    if (lo > hi) { const double t = lo; lo = hi; hi = t; }
    if (hi - lo > 1.4901161193847656e-08) { S = hi - lo; c = lo; }
}

// value descriptors of the descent program (shared pattern): what a CSC entry of A / G or an entry of b / h holds
enum { SG_A_DIAG = 0, SG_A_NEGA = 1, SG_A_NEGBM = 2, SG_A_NEGBP = 3, SG_A_ZERO = 4 };
struct SgProg {
    int n, p, m, l, nnzA, nnzG, N2;
    const int *a_kind, *a_i, *a_j;        // per nnz of A: kind, (row, col) inside the 4 x 4 / 4 x 2 block
    const int *g_kind;                    // per nnz of G: 0 constant (g_val), 1 = -Sx[1]
    const double* g_val;
    const int* b_kind;                    // per row of A: 0 x0 row, 1 xf row, 2 dynamics row; b_i = component
    const int* b_i;
    const int* h_kind;                    // per row of G: 0 constant (h_val), 1 = cx[1]
    const double* h_val;
    const double* lti;                    // [NCAND][36]: A (row-major 4x4), Bm (4x2), Bp (4x2), r (4)
    // row equilibration (round 5, see oracle/starship_guess.py): every row / cone divided by its largest |coefficient|
    const int *a_row, *g_row;             // row of every CSC entry of A / G
    const int* gs_kind;                   // per row of G: 0 = constant factor gs_val, 1 = 1 / Sx[1] (the ground rows)
    const double* gs_val;
    double Su[2], cu[2], vf[2];
};
// 1 / (largest |coefficient|) of equality row (kind, component i) of the descent program: boundary rows hold Sx[i] only, a
// dynamics row holds Sx[i], -A[i][:] Sx, -Bm[i][:] Su, -Bp[i][:] Su
__device__ __forceinline__ double sg_eq_row_scale(int kind, int i, const double* Sx, const double* Su, const double* Am, const double* Bm, const double* Bp)
{
    double mx = fabs(Sx[i]);
    if (kind == 2) {
        for (int j = 0; j < 4; j++) mx = fmax(mx, fabs(Am[i * 4 + j] * Sx[j]));
        for (int j = 0; j < 2; j++) mx = fmax(mx, fmax(fabs(Bm[i * 2 + j] * Su[j]), fabs(Bp[i * 2 + j] * Su[j])));
    }
    return mx > 0.0 ? 1.0 / mx : 1.0;
}
struct SgFill {
    int B;                                // instances in this chunk
    long BS;
    const double* xs;                     // [B][8]
    const int* ok1;
    double *c, *b, *h, *Gx, *Ax;          // engine inputs, interleaved [len][BS], problem t = inst * NCAND + cand
    int* active;                          // [B * NCAND]
};

__global__ void starship_descent_fill_kernel(SgFill f, SgProg P)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)f.B * SG_NCAND) return;
    const int inst = (int)(t / SG_NCAND), cand = (int)(t % SG_NCAND);
    f.active[t] = f.ok1[inst];
    if (!f.ok1[inst]) return;
    const double* xs = f.xs + (long)inst * 8;
    double Sx[4], cx[4];
    for (int i = 0; i < 4; i++) sg_scale(0.0, xs[i], Sx[i], cx[i]);
    const double* L = P.lti + (long)cand * 36;
    const double *Am = L, *Bm = L + 16, *Bp = L + 24, *r = L + 32;
    for (int e = 0; e < P.n; e++) f.c[(long)e * f.BS + t] = 0.0;
    for (int e = 0; e < P.nnzA; e++) {
        const int i = P.a_i[e], j = P.a_j[e];
        double v = 0.0;
        switch (P.a_kind[e]) {
            case SG_A_DIAG: v = i == j ? Sx[j] : 0.0; break;
            case SG_A_NEGA: v = -Am[i * 4 + j] * Sx[j]; break;
            case SG_A_NEGBM: v = -Bm[i * 2 + j] * P.Su[j]; break;
            case SG_A_NEGBP: v = -Bp[i * 2 + j] * P.Su[j]; break;
            default: v = 0.0;
        }
        const int rw = P.a_row[e];
        f.Ax[(long)e * f.BS + t] = v * sg_eq_row_scale(P.b_kind[rw], P.b_i[rw], Sx, P.Su, Am, Bm, Bp);
    }
    const double xf[4] = {0.0, 0.0, P.vf[0], P.vf[1]};
    for (int rw = 0; rw < P.p; rw++) {
        const int i = P.b_i[rw];
        double v;
        if (P.b_kind[rw] == 0) v = -(cx[i] - xs[i]);
        else if (P.b_kind[rw] == 1) v = -(cx[i] - xf[i]);
        else {
            double acc = cx[i];
            for (int q = 0; q < 4; q++) acc -= Am[i * 4 + q] * cx[q];
            for (int q = 0; q < 2; q++) acc -= Bm[i * 2 + q] * P.cu[q];
            for (int q = 0; q < 2; q++) acc -= Bp[i * 2 + q] * P.cu[q];
            acc -= r[i];
            v = -acc;
        }
        f.b[(long)rw * f.BS + t] = v * sg_eq_row_scale(P.b_kind[rw], i, Sx, P.Su, Am, Bm, Bp);
    }
    const double ground = fabs(Sx[1]) > 0.0 ? 1.0 / fabs(Sx[1]) : 1.0;
    for (int e = 0; e < P.nnzG; e++) {
        const int rw = P.g_row[e];
        f.Gx[(long)e * f.BS + t] = (P.g_kind[e] == 0 ? P.g_val[e] : -Sx[1]) * (P.gs_kind[rw] == 0 ? P.gs_val[rw] : ground);
    }
    for (int rw = 0; rw < P.m; rw++)
        f.h[(long)rw * f.BS + t] = (P.h_kind[rw] == 0 ? P.h_val[rw] : cx[1]) * (P.gs_kind[rw] == 0 ? P.gs_val[rw] : ground);
}

struct SgRec {
    int B, N, n1, N2, id_sw;
    long BS;
    const double* z;          // engine solution x, interleaved
    const int* status;        // [B * NCAND]
    const double *xs, *t1;
    const int* ok1;
    double *xd, *ud, *p;
    int* fail;                // [B]: 1 when the reference guess could not be built (phase 1 or no feasible duration)
    double Su[2], cu[2];
    double tau_s, alpha_e;
};

__global__ void starship_reconstruct_kernel(SgRec a)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    a.fail[b] = 1;
    if (!a.ok1[b]) return;
    int cand = -1;
    for (int c = 0; c < SG_NCAND && cand < 0; c++) if (a.status[(long)b * SG_NCAND + c] <= 1) cand = c;
    if (cand < 0) return;       // "could not find a terminal descent time of flight" (:415-419)
    a.fail[b] = 0;
    const long t = (long)b * SG_NCAND + cand;
    const double t2 = 10.0 + (double)cand;
    const double* xs = a.xs + (long)b * 8;
    double Sx[4], cx[4];
    for (int i = 0; i < 4; i++) sg_scale(0.0, xs[i], Sx[i], cx[i]);
    double* xo = a.xd + (long)b * a.N * 8;
    double* uo = a.ud + (long)b * a.N * 3;
    const double tdil = t2 / (1.0 - a.tau_s);
    const double tau0 = sg_linrange(0.0, 1.0, a.N, a.id_sw);
    const double m20 = xo[(long)a.id_sw * 8 + 6];
    double mass_acc = 0.0, th_prev = 0.0, tt_prev = 0.0, ff_prev = 0.0;
    for (int k = 0; k < a.N2; k++) {
        const int j = a.id_sw + k;
        double X2[4], T2[2];
        for (int i = 0; i < 4; i++) X2[i] = Sx[i] * a.z[(long)(4 * k + i) * a.BS + t] + cx[i];
        for (int i = 0; i < 2; i++) T2[i] = a.Su[i] * a.z[(long)(4 * a.N2 + 2 * k + i) * a.BS + t] + a.cu[i];
        for (int i = 0; i < 4; i++) xo[(long)j * 8 + i] = X2[i];
        const double th = -atan2(T2[0], T2[1]);
        const double Tn = sqrt(T2[0] * T2[0] + T2[1] * T2[1]);
        xo[(long)j * 8 + 4] = th;
        uo[(long)j * 3 + 0] = Tn;
        const double tau2 = sg_linrange(0.0, 1.0, a.N, j) - tau0;
        const double tt = tau2 * tdil, ff = a.alpha_e * Tn;
        if (k > 0) {
            const double tau2p = sg_linrange(0.0, 1.0, a.N, j - 1) - tau0;
            xo[(long)(j - 1) * 8 + 5] = (th - th_prev) / ((tau2 - tau2p) * tdil);
            mass_acc += 0.5 * (tt - tt_prev) * (ff + ff_prev);
            xo[(long)j * 8 + 6] = m20 + mass_acc;
        }
        th_prev = th; tt_prev = tt; ff_prev = ff;
    }
    double* po = a.p + (long)b * 10;
    po[0] = a.t1[b]; po[1] = t2;
    for (int i = 0; i < 8; i++) po[2 + i] = xs[i];
}

// ---------------- host side: the candidate FOH models and the shared pattern ----------------
// FOH discretisation of the descent double integrator over one normalised interval, the reference's own RK4 recipe on
// LinRange(0, dt, 100) with V = [Phi; int iPhi B s-; int iPhi B s+; int iPhi r] (definition.jl:183-231)
inline void sg_descent_lti(double dtn, double tdil, double m, double g0, double (&out)[36])
{
    double A[16] = {0}, Bc[8] = {0}, r[4] = {0, 0, 0, -g0 * tdil};
    A[0 * 4 + 2] = tdil; A[1 * 4 + 3] = tdil;                 // row-major
    Bc[2 * 2 + 0] = tdil / m; Bc[3 * 2 + 1] = tdil / m;
    auto inv4 = [](const double* Min, double* Mout) {        // Gauss-Jordan with partial pivoting
        double a[4][8];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = Min[i * 4 + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
        for (int c = 0; c < 4; c++) {
            int pv = c;
            for (int i = c + 1; i < 4; i++) if (std::fabs(a[i][c]) > std::fabs(a[pv][c])) pv = i;
            if (pv != c) for (int j = 0; j < 8; j++) std::swap(a[c][j], a[pv][j]);
            const double d = a[c][c];
            for (int j = 0; j < 8; j++) a[c][j] /= d;
            for (int i = 0; i < 4; i++) if (i != c) { const double f = a[i][c]; for (int j = 0; j < 8; j++) a[i][j] -= f * a[c][j]; }
        }
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Mout[i * 4 + j] = a[i][4 + j];
    };
    auto derivs = [&](double t, const double* V, double* dV) {
        const double* Phi = V;                                // row-major 4x4
        const double sm = (dtn - t) / dtn, sp = t / dtn;
        double iPhi[16];
        inv4(Phi, iPhi);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double acc = 0; for (int q = 0; q < 4; q++) acc += A[i * 4 + q] * Phi[q * 4 + j]; dV[i * 4 + j] = acc; }
        for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) {
            double acc = 0; for (int q = 0; q < 4; q++) acc += iPhi[i * 4 + q] * Bc[q * 2 + j];
            dV[16 + i * 2 + j] = acc * sm; dV[24 + i * 2 + j] = acc * sp;
        }
        for (int i = 0; i < 4; i++) { double acc = 0; for (int q = 0; q < 4; q++) acc += iPhi[i * 4 + q] * r[q]; dV[32 + i] = acc; }
    };
    double V[36] = {0};
    for (int i = 0; i < 4; i++) V[i * 4 + i] = 1.0;
    for (int j = 1; j < 100; j++) {
        const double ta = (1.0 - (double)(j - 1) / 99.0) * 0.0 + ((double)(j - 1) / 99.0) * dtn;
        const double tb = (1.0 - (double)j / 99.0) * 0.0 + ((double)j / 99.0) * dtn;
        const double h = tb - ta;
        double k1[36], k2[36], k3[36], k4[36], Y[36];
        derivs(ta, V, k1);
        for (int i = 0; i < 36; i++) Y[i] = V[i] + h / 2 * k1[i];
        derivs(ta + h / 2, Y, k2);
        for (int i = 0; i < 36; i++) Y[i] = V[i] + h / 2 * k2[i];
        derivs(ta + h / 2, Y, k3);
        for (int i = 0; i < 36; i++) Y[i] = V[i] + h * k3[i];
        derivs(ta + h, Y, k4);
        for (int i = 0; i < 36; i++) V[i] = V[i] + h / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    }
    const double* Ak = V;
    for (int i = 0; i < 16; i++) out[i] = Ak[i];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) {
        double am = 0, ap = 0;
        for (int q = 0; q < 4; q++) { am += Ak[i * 4 + q] * V[16 + q * 2 + j]; ap += Ak[i * 4 + q] * V[24 + q * 2 + j]; }
        out[16 + i * 2 + j] = am; out[24 + i * 2 + j] = ap;
    }
    for (int i = 0; i < 4; i++) { double acc = 0; for (int q = 0; q < 4; q++) acc += Ak[i * 4 + q] * V[32 + q]; out[32 + i] = acc; }
}

// pattern of the descent program in CSC (sorted rows per column) + the per-entry descriptors
struct SgPattern {
    int n = 0, p = 0, m = 0, l = 0, N2 = 0;
    conic::Csc A, G;
    std::vector<int> q;
    std::vector<int> a_kind, a_i, a_j, g_kind, b_kind, b_i, h_kind;
    std::vector<double> g_val, h_val;
    std::vector<int> gs_kind;          // row scale of G: 0 constant (gs_val), 1 = 1 / Sx[1]
    std::vector<double> gs_val;
};
inline SgPattern sg_build_pattern(int N2, const double Su[2], const double cu[2], double T_min1, double T_max1, double theta_max2)
{
    SgPattern S;
    S.N2 = N2; S.n = 6 * N2;
    struct Ent { int r, c, kind, i, j; double v; };
    std::vector<Ent> ea, eg;
    auto ix = [&](int k, int i) { return 4 * k + i; };
    auto iu = [&](int k, int i) { return 4 * N2 + 2 * k + i; };
    int nr = 0;
    auto block = [&](int row0, int col0, int nc, int kind) { for (int i = 0; i < 4; i++) for (int j = 0; j < nc; j++) ea.push_back({row0 + i, col0 + j, kind, i, j, 0.0}); };
    block(nr, ix(0, 0), 4, SG_A_DIAG); for (int i = 0; i < 4; i++) { S.b_kind.push_back(0); S.b_i.push_back(i); } nr += 4;
    block(nr, ix(N2 - 1, 0), 4, SG_A_DIAG); for (int i = 0; i < 4; i++) { S.b_kind.push_back(1); S.b_i.push_back(i); } nr += 4;
    for (int k = 0; k < N2 - 1; k++) {
        block(nr, ix(k + 1, 0), 4, SG_A_DIAG); block(nr, ix(k, 0), 4, SG_A_NEGA); block(nr, iu(k, 0), 2, SG_A_NEGBM); block(nr, iu(k + 1, 0), 2, SG_A_NEGBP);
        for (int i = 0; i < 4; i++) { S.b_kind.push_back(2); S.b_i.push_back(i); }
        nr += 4;
    }
    S.p = nr;
    int ng = 0;
    for (int k = 0; k < N2; k++) {   // T_min1 - u_y <= 0 ; -r_y <= 0
        eg.push_back({ng, iu(k, 1), 0, 0, 0, -Su[1]}); S.h_kind.push_back(0); S.h_val.push_back(-(T_min1 - cu[1])); ng++;
        S.gs_kind.push_back(0); S.gs_val.push_back(1.0 / std::fabs(Su[1]));
        eg.push_back({ng, ix(k, 1), 1, 0, 0, 0.0}); S.h_kind.push_back(1); S.h_val.push_back(0.0); ng++;
        S.gs_kind.push_back(1); S.gs_val.push_back(1.0);
    }
    S.l = ng;
    const double ct = 1.0 / std::cos(theta_max2);
    for (int k = 0; k < N2; k++) {
        // (T_max1, u) in Q^3
        S.h_kind.insert(S.h_kind.end(), {0, 0, 0}); S.h_val.insert(S.h_val.end(), {T_max1, cu[0], cu[1]});
        eg.push_back({ng + 1, iu(k, 0), 0, 0, 0, -Su[0]}); eg.push_back({ng + 2, iu(k, 1), 0, 0, 0, -Su[1]});
        ng += 3; S.q.push_back(3);
        for (int r3 = 0; r3 < 3; r3++) { S.gs_kind.push_back(0); S.gs_val.push_back(1.0 / std::fmax(std::fabs(Su[0]), std::fabs(Su[1]))); }   // one factor per cone
        // (u_y / cos(theta_max2), u) in Q^3
        S.h_kind.insert(S.h_kind.end(), {0, 0, 0}); S.h_val.insert(S.h_val.end(), {cu[1] * ct, cu[0], cu[1]});
        eg.push_back({ng, iu(k, 1), 0, 0, 0, -Su[1] * ct}); eg.push_back({ng + 1, iu(k, 0), 0, 0, 0, -Su[0]}); eg.push_back({ng + 2, iu(k, 1), 0, 0, 0, -Su[1]});
        ng += 3; S.q.push_back(3);
        for (int r3 = 0; r3 < 3; r3++) { S.gs_kind.push_back(0); S.gs_val.push_back(1.0 / std::fmax(std::fabs(Su[1]) * ct, std::fabs(Su[0]))); }
    }
    S.m = ng;
    auto to_csc = [&](std::vector<Ent>& e, int nrow, conic::Csc& M, auto&& emit) {
        std::stable_sort(e.begin(), e.end(), [](const Ent& a, const Ent& b) { return a.c != b.c ? a.c < b.c : a.r < b.r; });
        M.nrow = nrow; M.ncol = S.n; M.p.assign(S.n + 1, 0); M.i.clear();
        for (const Ent& x : e) { M.p[x.c + 1]++; M.i.push_back(x.r); emit(x); }
        for (int c = 0; c < S.n; c++) M.p[c + 1] += M.p[c];
    };
    to_csc(ea, S.p, S.A, [&](const Ent& x) { S.a_kind.push_back(x.kind); S.a_i.push_back(x.i); S.a_j.push_back(x.j); });
    to_csc(eg, S.m, S.G, [&](const Ent& x) { S.g_kind.push_back(x.kind); S.g_val.push_back(x.v); });
    return S;
}

}  // namespace scp
