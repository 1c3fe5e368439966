// Starship landing flip, test/examples/starship_flip/{parameters,definition}.jl:
// x = [r(2); v(2); theta; omega; m; delta_d], u = [T; delta; delta_dot], p = [t1; t2; xs(8)]  (definition.jl:45-47,
// parameters.jl:110-123).  Two-phase time dilation (t <= tau_s: p[t1]/tau_s, else p[t2]/(1 - tau_s), definition.jl:515),
// thrust vector, quadratic drag, aerodynamic and thrust torques, first-order gimbal delay (:498-550); Jacobians
// :552-637.  The first model with STATE-DEPENDENT Jacobians: discretize! runs the reference-form kernel K1 (cooperative
// LU per stage), and the subproblem goes through the generic conic path (np = 10, 21 non-convex rows).
#pragma once
#include "model_common.hpp"

namespace scp {

struct Starship : ModelDefaults {
    static constexpr int id = 3;
    static constexpr int nx = 8, nu = 3, np = 10, npF = 2;   // F: only the columns of t1 and t2 are ever non-zero (:627-634)
    static constexpr bool const_jacobian = false;
    static constexpr double var_form_max_step = 0.0;
    static constexpr bool has_subproblem = true;   // false: discretize! / propagate / guess only (freeflyer.hpp)
    static constexpr bool structured = false;                // no stage-structured fast path (np = 10, ns = 21)
    // Parameter blob (include/scp_mi355x.h, scp_problem_desc.model_par) -- everything the reference keeps in `traj.mdl`
    // (parameters.jl:99-212) is data:
    //   [N, hs, g0, m, lcg, lcp, J, CD, T_min1, T_max1, T_min3, T_max3, alpha_e, delta_max, deltadot_max, rate_delay,
    //    tf_min, tf_max, tau_s, gamma_gs (rad), theta_max2, vf_x, vf_y, cost weight of the switch altitude, mass scale of the cost]
    // N: s(.) needs the grid to find the phase-switch node (:709); hs = altitude normalisation of the cost, which the
    // reference's guess generator overwrites (:181).
    static constexpr int npar = 25;
    struct Params {
        int N;
        double g0, m, lcg, lcp, J, CD;
        double T_min1, T_max1, T_min3, T_max3;
        double alpha_e, delta_max, deltadot_max, rate_delay;
        double tf_min, tf_max, tau_s, hs;
        double cos_gs, theta_max2;
        double vf_x, vf_y;
        double cost_alt, cost_mass;      // terminal cost cost_alt * (-alt(xs) / hs) + (0 - m_N) / cost_mass (:456-476)
    };
    static Params make_params(const double* par)
    {
        Params P;
        P.N = (int)par[0];
        P.hs = par[1];
        P.g0 = par[2]; P.m = par[3]; P.lcg = par[4]; P.lcp = par[5]; P.J = par[6]; P.CD = par[7];
        P.T_min1 = par[8]; P.T_max1 = par[9]; P.T_min3 = par[10]; P.T_max3 = par[11];
        P.alpha_e = par[12]; P.delta_max = par[13]; P.deltadot_max = par[14]; P.rate_delay = par[15];
        P.tf_min = par[16]; P.tf_max = par[17]; P.tau_s = par[18];
        P.cos_gs = cos(par[19]); P.theta_max2 = par[20];
        P.vf_x = par[21]; P.vf_y = par[22];
        P.cost_alt = par[23]; P.cost_mass = par[24];
        return P;
    }
    static constexpr int Fcol(int j) { return j; }

    SCP_DEV static double tdil(const Params& P, double t, const double* p) { return t <= P.tau_s ? p[0] / P.tau_s : p[1] / (1.0 - P.tau_s); }

    // f, A (col-major nx*nx), B (nx*nu), Fc (nx*npF: columns of t1, t2); T = double (reference arithmetic) or float
    // (the fp32 tolerance-check variant of K1: every operation below is carried out in T)
    static constexpr bool has_fp32 = true;
    // nothing depends on the position: A[:, r] = 0 (:552-586), Phi[:, r] = e_r; the r rows do see the velocity (A[r, v] = tdil I)
    static constexpr int lu_lead = 2;
    template <class T>
    SCP_DEV static void dyn(const Params& P, T t, int, const T (&x)[nx], const T (&u)[nu], const double* p,
                            T (&f)[nx], T (&A)[nx * nx], T (&B)[nx * nu], T (&Fc)[nx * npF])
    {
        const T one = (T)1, nil = (T)0;
        const T m = (T)P.m, J = (T)P.J, CD = (T)P.CD, g0 = (T)P.g0, alpha_e = (T)P.alpha_e, rate = (T)P.rate_delay, tau_s = (T)P.tau_s;
        const T vx = x[2], vy = x[3], th = x[4], om = x[5], dd = x[7];
        const T Th = u[0], de = u[1];
        const T pt[2] = {(T)p[0], (T)p[1]};
        const T td = t <= tau_s ? pt[0] / tau_s : pt[1] / (one - tau_s);
        const T leng = -(T)P.lcg, lcp = (T)P.lcp - (T)P.lcg;
        const T c = cos(th), s = sin(th), cd = cos(de), sd = sin(de);
        const T ei[2] = {c, s}, ej[2] = {-s, c};
        const T nv = sqrt(vx * vx + vy * vy);
        const T Tv[2] = {Th * (-sd * ei[0] + cd * ej[0]), Th * (-sd * ei[1] + cd * ej[1])};
        const T MT = leng * Th * sd;
        const T D[2] = {-CD * nv * vx, -CD * nv * vy};
        const T MD = -lcp * (D[0] * ei[0] + D[1] * ei[1]);
        f[0] = vx; f[1] = vy;
        f[2] = (Tv[0] + D[0]) / m; f[3] = (Tv[1] + D[1]) / m - g0;
        f[4] = om; f[5] = (MT + MD) / J; f[6] = alpha_e * Th; f[7] = (de - dd) / rate;
#pragma unroll
        for (int i = 0; i < nx; i++) f[i] *= td;
        // ---- A (:552-586) ----
#pragma unroll
        for (int i = 0; i < nx * nx; i++) A[i] = nil;
        const T inv = nv > nil ? one / nv : nil;
        const T gD[2][2] = {{-CD * (nv + vx * vx * inv), -CD * (vx * vy * inv)},      // grad_v D (symmetric)
                            {-CD * (vx * vy * inv), -CD * (nv + vy * vy * inv)}};
        const T gthTv[2] = {Th * (-sd * ej[0] - cd * ei[0]), Th * (-sd * ej[1] - cd * ei[1])};
        const T gvMD[2] = {-lcp * (gD[0][0] * ei[0] + gD[1][0] * ei[1]), -lcp * (gD[0][1] * ei[0] + gD[1][1] * ei[1])};
        const T gthMD = -lcp * (D[0] * ej[0] + D[1] * ej[1]);
        A[0 + nx * 2] = one; A[1 + nx * 3] = one;
        A[2 + nx * 2] = gD[0][0] / m; A[2 + nx * 3] = gD[0][1] / m; A[3 + nx * 2] = gD[1][0] / m; A[3 + nx * 3] = gD[1][1] / m;
        A[2 + nx * 4] = gthTv[0] / m; A[3 + nx * 4] = gthTv[1] / m;
        A[4 + nx * 5] = one;
        A[5 + nx * 2] = gvMD[0] / J; A[5 + nx * 3] = gvMD[1] / J; A[5 + nx * 4] = gthMD / J;
        A[7 + nx * 7] = -one / rate;
#pragma unroll
        for (int i = 0; i < nx * nx; i++) A[i] *= td;
        // ---- B (:587-624) ----
#pragma unroll
        for (int i = 0; i < nx * nu; i++) B[i] = nil;
        B[2 + nx * 0] = (-sd * ei[0] + cd * ej[0]) / m; B[3 + nx * 0] = (-sd * ei[1] + cd * ej[1]) / m;
        B[2 + nx * 1] = Th * (-cd * ei[0] - sd * ej[0]) / m; B[3 + nx * 1] = Th * (-cd * ei[1] - sd * ej[1]) / m;
        B[5 + nx * 0] = leng * sd / J; B[5 + nx * 1] = leng * Th * cd / J;
        B[6 + nx * 0] = alpha_e; B[7 + nx * 1] = one / rate;
#pragma unroll
        for (int i = 0; i < nx * nu; i++) B[i] *= td;
        // ---- F (:625-636): F[:, id_t] = f / p[id_t] ----
        const int it = t <= tau_s ? 0 : 1;
#pragma unroll
        for (int i = 0; i < nx; i++) { Fc[i + nx * it] = f[i] / pt[it]; Fc[i + nx * (1 - it)] = nil; }
    }
    // the variational kernel is never selected for this model (const_jacobian = false); stubs keep the templates complete
    SCP_DEV static void Amul(const Params&, const double*, const double (&)[nx], double (&out)[nx]) { zero(out); }
    SCP_DEV static void Bcol(const Params&, const double*, int, double (&out)[nx]) { zero(out); }
    template <class T>
    SCP_DEV static void action(T (&)[nx]) {}
    static constexpr bool has_impulse = false;
    SCP_DEV static void impulse(const Params&, double, int, const double (&)[nx], const double (&)[nu], const double*,
                                double (&dx)[nx], double (&B)[nx * nu])
    {
        zero(dx); zero(B);
    }
    // Straight-line guess between the boundary states (the reference's bang-bang + LCvx guess, definition.jl:97-445, is
    // host-side pre-processing and can be passed in as a warm start): x from pp to the landing state, hover thrust,
    // t1 = t2 = 10 s, xs = state at the phase switch.
    SCP_DEV static void guess(const Params& P, const double* pp, int N, int k, double (&x)[nx], double (&u)[nu], double* p, double*)
    {
        const double t = (double)k / (double)(N - 1);
        const double x0[nx] = {pp[0], pp[1], pp[2], pp[3], pp[4], 0.0, 0.0, 0.0};
        const double xf[nx] = {0.0, 0.0, P.vf_x, P.vf_y, 0.0, 0.0, -3e3, 0.0};
#pragma unroll
        for (int i = 0; i < nx; i++) x[i] = (1.0 - t) * x0[i] + t * xf[i];
        u[0] = t <= P.tau_s ? P.T_min3 : P.m * P.g0; u[1] = 0.0; u[2] = 0.0;
        if (k == 0) {
            p[0] = 10.0; p[1] = 10.0;
#pragma unroll
            for (int i = 0; i < nx; i++) p[2 + i] = 0.5 * (x0[i] + xf[i]);
        }
    }

    // ---- subproblem side ----
    static constexpr int ns = 7 + 2 * nx, nl = 5, nsoc = 0, ng = 2, nic = 7, ntc = 6, npp = 5;  // pp = [r0(2) v0(2) theta0]
    SCP_DEV static bool phase_switch(const Params& P, double t)
    {
        const double dt = 1.0 / (double)(P.N - 1), tol = 1e-3;
        return (P.tau_s - dt) + tol <= t && t <= P.tau_s + tol;      // definition.jl:705-712
    }
    SCP_DEV static bool phase2(const Params& P, double t) { return phase_switch(P, t) || t > P.tau_s; }
    // s (21 rows), C, D, G (row-major) -- definition.jl:723-810
    SCP_DEV static void s_eval(const Params& P, double t, int, const double* x, const double* u, const double* p, double* s,
                               double* C, double* Dm, double* G)
    {
        for (int i = 0; i < ns; i++) s[i] = 0.0;
        for (int i = 0; i < ns * nx; i++) C[i] = 0.0;
        for (int i = 0; i < ns * nu; i++) Dm[i] = 0.0;
        for (int i = 0; i < ns * np; i++) G[i] = 0.0;
        const double dd = x[7], de = u[1], dedot = u[2];
        s[0] = (de - dd) - dedot * P.rate_delay;
        s[1] = dedot * P.rate_delay - (de - dd);
        s[2] = dedot - P.deltadot_max;
        s[3] = -P.deltadot_max - dedot;
        const double nr = sqrt(x[0] * x[0] + x[1] * x[1]);
        s[4] = nr * P.cos_gs - x[1];
        C[0 * nx + 7] = -1.0; C[1 * nx + 7] = 1.0;
        const bool tiny = nr < 1.4901161193847656e-08;   // sqrt(eps), :757
        C[4 * nx + 0] = (tiny ? 0.0 : x[0] / nr) * P.cos_gs;
        C[4 * nx + 1] = (tiny ? 0.0 : x[1] / nr) * P.cos_gs - 1.0;
        Dm[0 * nu + 1] = 1.0; Dm[0 * nu + 2] = -P.rate_delay; Dm[1 * nu + 1] = -1.0; Dm[1 * nu + 2] = P.rate_delay;
        Dm[2 * nu + 2] = 1.0; Dm[3 * nu + 2] = -1.0;
        if (phase_switch(P, t)) {
            for (int i = 0; i < nx; i++) {
                s[5 + i] = p[2 + i] - x[i];
                s[5 + nx + i] = x[i] - p[2 + i];
                C[(5 + i) * nx + i] = -1.0; C[(5 + nx + i) * nx + i] = 1.0;
                G[(5 + i) * np + 2 + i] = 1.0; G[(5 + nx + i) * np + 2 + i] = -1.0;
            }
        }
        if (phase2(P, t)) {
            s[ns - 2] = x[4] - P.theta_max2;
            s[ns - 1] = -P.theta_max2 - x[4];
            C[(ns - 2) * nx + 4] = 1.0; C[(ns - 1) * nx + 4] = -1.0;
        }
    }
    // X: v_y <= 0 (:649-652); U: T_min(t) <= T <= T_max(t), |delta| <= delta_max (the reference's 1-dimensional L1 cone,
    // :686-698) -- rows over z = [x; u]
    SCP_DEV static void lin_rows(const Params& P, double t, int, double* L, double* Lp, double* l)
    {
        constexpr int nz = nx + nu;
        for (int i = 0; i < nl * nz; i++) L[i] = 0.0;
        for (int i = 0; i < nl * np; i++) Lp[i] = 0.0;
        const bool flip = t <= P.tau_s;
        L[0 * nz + 3] = 1.0; l[0] = 0.0;
        L[1 * nz + nx + 0] = 1.0; l[1] = -(flip ? P.T_max3 : P.T_max1);
        L[2 * nz + nx + 0] = -1.0; l[2] = flip ? P.T_min3 : P.T_min1;
        L[3 * nz + nx + 1] = 1.0; l[3] = -P.delta_max;
        L[4 * nz + nx + 1] = -1.0; l[4] = -P.delta_max;
    }
    SCP_DEV static void soc_rows(const Params&, double, int, double*, double*) {}
    // tf_min <= t1 + t2 <= tf_max (:653-668; repeated at every node by the reference, kept once)
    SCP_DEV static void glin_rows(const Params& P, double* Lp, double* lp)
    {
        for (int i = 0; i < ng * np; i++) Lp[i] = 0.0;
        Lp[0 * np + 0] = 1.0; Lp[0 * np + 1] = 1.0; lp[0] = -P.tf_max;
        Lp[1 * np + 0] = -1.0; Lp[1 * np + 1] = -1.0; lp[1] = P.tf_min;
    }
    // ic: [r; v; theta; omega; m] = [r0; v0; theta0; 0; 0] (:814-842); tc: [r; v; theta; omega] = [0; vf; 0; 0] (:843-870)
    SCP_DEV static void bc_ic(const Params&, const double* x, const double*, const double* pp, double* g, double* H, double* K)
    {
        for (int i = 0; i < nic * nx; i++) H[i] = 0.0;
        for (int i = 0; i < nic * np; i++) K[i] = 0.0;
        const double rhs[nic] = {pp[0], pp[1], pp[2], pp[3], pp[4], 0.0, 0.0};
        for (int i = 0; i < nic; i++) { g[i] = x[i] - rhs[i]; H[i * nx + i] = 1.0; }
    }
    SCP_DEV static void bc_tc(const Params& P, const double* x, const double*, const double*, double* g, double* H, double* K)
    {
        for (int i = 0; i < ntc * nx; i++) H[i] = 0.0;
        for (int i = 0; i < ntc * np; i++) K[i] = 0.0;
        const double rhs[ntc] = {0.0, 0.0, P.vf_x, P.vf_y, 0.0, 0.0};
        for (int i = 0; i < ntc; i++) { g[i] = x[i] - rhs[i]; H[i * nx + i] = 1.0; }
    }
    // terminal cost mu * (-alt(xs) / hs) + (0 - m_N) / 10e3, mu = 0.3 (:456-476)
    SCP_DEV static void cost_terms(const Params& P, double* Qu, double* lu, double* lx, double* tx, double* tp, double* Qp)
    {
        for (int i = 0; i < nu; i++) { Qu[i] = 0.0; lu[i] = 0.0; }
        for (int i = 0; i < nx; i++) { lx[i] = 0.0; tx[i] = 0.0; }
        for (int i = 0; i < np; i++) { tp[i] = 0.0; Qp[i] = 0.0; }
        tx[6] = -1.0 / P.cost_mass;
        tp[2 + 1] = -P.cost_alt / P.hs;
    }
};

}  // namespace scp
