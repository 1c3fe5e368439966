// 6-DoF free-flyer in the space station, test/examples/freeflyer/{parameters,definition}.jl:
// x = [r(3); v(3); q(4); w(3)] (quaternion q = [vector; scalar], src/utils/quaternion.jl:33-36), u = [T(3); M(3)],
// dynamics r' = v, v' = T/m, q' = 1/2 q (x) w, w' = J^-1 (M - w x J w), all times the time dilation p[0]
// (definition.jl:224-284), integration action q <- q/|q| after every RK4 step (:69-82) -- the first model with a real
// `action` and with a 13-dimensional state-dependent Jacobian.
//
// SCOPE: discretize!, propagate and the initial guess.  The reference's parameter vector is p = [t_f; delta] with one
// room-SDF slack per room and node (np = 1 + 6N, parameters.jl:121-128); the slacks never enter the dynamics (F has the
// single structurally non-zero column of t_f), so the compiled model carries np = 1 and discretize! is exact.  The
// SUBPROBLEM of this model needs delta (X rows and the logsumexp row of s, definition.jl:286-349, 381-452), i.e. a
// parameter count that depends on N: has_subproblem = false, the subproblem entry points refuse the model.
#pragma once
#include "model_common.hpp"

namespace scp {

struct Freeflyer {
    static constexpr int id = 4;
    static constexpr int nx = 13, nu = 6, np = 1, npF = 1;
    static constexpr bool const_jacobian = false;
    static constexpr double var_form_max_step = 0.0;
    static constexpr bool structured = false;
    static constexpr bool has_subproblem = false;
    static constexpr int npar = 4;  // [m, J1, J2, J3] (parameters.jl:140-141)

    struct Params {
        double m, J[3];
        // parameters.jl:135-139, 162-166
        double v_max = 0.4, w_max = 3.14159265358979323846 / 180.0, T_max = 20e-3, M_max = 1e-4;
        double tf_min = 60.0, tf_max = 200.0, gamma = 0.0;
        // obstacles (parameters.jl:95-101): H = I / 0.3
        double obs_h = 1.0 / 0.3;
        double obs_c[3][3] = {{8.5, -0.15, 5.0}, {11.2, 1.84, 5.0}, {11.3, 3.8, 4.8}};
    };
    static Params make_params(const double* par)
    {
        Params P;
        P.m = par[0]; P.J[0] = par[1]; P.J[1] = par[2]; P.J[2] = par[3];
        return P;
    }
    static constexpr int Fcol(int) { return 0; }

    SCP_DEV static void cross(const double* a, const double* b, double* c)
    {
        c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
    }
    // f, A (col-major nx*nx), B (nx*nu), Fc (nx: the t_f column)
    SCP_DEV static void dyn(const Params& P, double, int, const double (&x)[nx], const double (&u)[nu], const double* p,
                            double (&f)[nx], double (&A)[nx * nx], double (&B)[nx * nu], double (&Fc)[nx])
    {
        const double td = p[0];
        const double qv[3] = {x[6], x[7], x[8]}, qw = x[9], w[3] = {x[10], x[11], x[12]};
        double qxw[3], Jw[3], wxJw[3];
        cross(qv, w, qxw);
#pragma unroll
        for (int i = 0; i < 3; i++) Jw[i] = P.J[i] * w[i];
        cross(w, Jw, wxJw);
        double f0[nx];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            f0[i] = x[3 + i];
            f0[3 + i] = u[i] / P.m;
            f0[6 + i] = 0.5 * (qw * w[i] + qxw[i]);                 // vector part of 1/2 q (x) [w; 0]
            f0[10 + i] = (u[3 + i] - wxJw[i]) / P.J[i];
        }
        f0[9] = -0.5 * (qv[0] * w[0] + qv[1] * w[1] + qv[2] * w[2]);
#pragma unroll
        for (int i = 0; i < nx; i++) { f[i] = f0[i] * td; Fc[i] = f0[i]; }   // F[:, t_f] = f / tdil  (:278)
        zero(A);
#pragma unroll
        for (int i = 0; i < 3; i++) A[i + nx * (3 + i)] = td;
        // [a]x entries: S(i, j) of helper.jl:65-70
        const double Sw[3][3] = {{0.0, -w[2], w[1]}, {w[2], 0.0, -w[0]}, {-w[1], w[0], 0.0}};
        const double Sq[3][3] = {{0.0, -qv[2], qv[1]}, {qv[2], 0.0, -qv[0]}, {-qv[1], qv[0], 0.0}};
        const double SJ[3][3] = {{0.0, -Jw[2], Jw[1]}, {Jw[2], 0.0, -Jw[0]}, {-Jw[1], Jw[0], 0.0}};
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                A[(6 + i) + nx * (6 + j)] = -0.5 * Sw[i][j] * td;                              // 1/2 skew(Quaternion(w), :R)
                A[(6 + i) + nx * (10 + j)] = 0.5 * ((i == j ? qw : 0.0) + Sq[i][j]) * td;      // 1/2 skew(q)[:, 1:3]
                A[(10 + i) + nx * (10 + j)] = -(Sw[i][j] * P.J[j] - SJ[i][j]) / P.J[i] * td;   // -J^-1 ([w]x J - [J w]x)
            }
            A[(6 + i) + nx * 9] = 0.5 * w[i] * td;
            A[9 + nx * (6 + i)] = -0.5 * w[i] * td;
            A[9 + nx * (10 + i)] = -0.5 * qv[i] * td;
        }
        zero(B);
#pragma unroll
        for (int i = 0; i < 3; i++) { B[(3 + i) + nx * i] = td / P.m; B[(10 + i) + nx * (3 + i)] = td / P.J[i]; }
    }
    SCP_DEV static void Amul(const Params&, const double*, const double (&)[nx], double (&out)[nx]) { zero(out); }
    SCP_DEV static void Bcol(const Params&, const double*, int, double (&out)[nx]) { zero(out); }
    // integration action (definition.jl:69-82): renormalise the quaternion after every full RK4 step
    template <class T>
    SCP_DEV static void action(T (&x)[nx])
    {
        const T n = sqrt(x[6] * x[6] + x[7] * x[7] + x[8] * x[8] + x[9] * x[9]);
#pragma unroll
        for (int i = 6; i < 10; i++) x[i] /= n;
    }
    static constexpr bool has_fp32 = false;
    static constexpr bool has_impulse = false;
    SCP_DEV static void impulse(const Params&, double, int, const double (&)[nx], const double (&)[nu], const double*,
                                double (&dx)[nx], double (&B)[nx * nu])
    {
        zero(dx); zero(B);
    }

    static constexpr int ns = 3, nl = 0, nsoc = 4, ng = 2, nic = 13, ntc = 13, npp = 26;  // pp = [r0 v0 q0 w0 rf vf qf wf]

    // q0' (x) q1 with q = [v; w]; Log(q) -> (angle, axis)  (quaternion.jl:211-214, 257-260, 277-282)
    SCP_DEV static void qmul(const double* a, const double* b, double* r)
    {
        double axb[3];
        cross(a, b, axb);
        for (int i = 0; i < 3; i++) r[i] = a[3] * b[i] + b[3] * a[i] + axb[i];
        r[3] = a[3] * b[3] - (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
    }
    // initial guess at node k of N (definition.jl:84-186): an axis-by-axis (L1) path at constant speed, SLERP attitude,
    // constant body rate, idle inputs, p = (tf_min + tf_max) / 2
    SCP_DEV static void guess(const Params& P, const double* pp, int N, int k, double (&x)[nx], double (&u)[nu], double* p)
    {
        const double* r0 = pp; const double* q0 = pp + 6; const double* rf = pp + 13; const double* qf = pp + 19;
        const double T = 0.5 * (P.tf_min + P.tf_max);
        const double l1 = fabs(rf[0] - r0[0]) + fabs(rf[1] - r0[1]) + fabs(rf[2] - r0[2]);
        const double speed = l1 / T;
        const double mix = (double)k / (double)(N - 1);
        const double tk = (1.0 - mix) * 0.0 + mix * T;                       // straightline_interpolate([0], [T], N)
        double cum[3], acc = 0.0;
        for (int i = 0; i < 3; i++) { acc += fabs(rf[i] - r0[i]) / speed; cum[i] = acc; }
        for (int i = 0; i < 3; i++) { x[i] = rf[i]; x[3 + i] = 0.0; }
        for (int i = 0; i < 3; i++) {
            if (tk <= cum[i]) {
                const double t0 = i > 0 ? cum[i - 1] : 0.0, t1 = cum[i];
                for (int j = 0; j < 3; j++) { x[j] = j < i ? rf[j] : r0[j]; x[3 + j] = 0.0; }
                const double tc = fmax(t0, fmin(t1, tk)), c = (t1 - tc) / (t1 - t0);  // linterp, helper.jl:107-118
                x[i] = c * r0[i] + (1.0 - c) * rf[i];
                x[3 + i] = speed * ((rf[i] - r0[i]) >= 0.0 ? 1.0 : -1.0);
                break;
            }
        }
        // slerp_interpolate(q0, qf, mix) (quaternion.jl:483-490)
        double q0c[4] = {-q0[0], -q0[1], -q0[2], q0[3]}, dq[4];
        qmul(q0c, qf, dq);
        double nv = sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2]);
        const double ang = 2.0 * atan2(nv, dq[3]);
        double ax[3] = {dq[0] / nv, dq[1] / nv, dq[2] / nv};
        const double ha = 0.5 * mix * ang;
        double dqt[4] = {ax[0] * sin(ha), ax[1] * sin(ha), ax[2] * sin(ha), cos(ha)}, qt[4];
        qmul(q0, dqt, qt);
        for (int i = 0; i < 4; i++) x[6 + i] = qt[i];
        // constant body rate: Log(qf * q0') / T (:160-163)
        double e[4];
        qmul(qf, q0c, e);
        nv = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        const double rang = 2.0 * atan2(nv, e[3]);
        for (int i = 0; i < 3; i++) x[10 + i] = rang / T * e[i] / nv;
        for (int i = 0; i < nu; i++) u[i] = 0.0;
        if (k == 0) p[0] = T;
    }

    // ---- the delta-free part of the problem definition (kept for completeness; has_subproblem = false) ----
    // obstacles: s_i = 1 - ||H (r - c_i)||  (definition.jl:381-398, ellipsoid.jl:99-118)
    SCP_DEV static void s_eval(const Params& P, double, int, const double* x, const double*, const double*, double* s,
                               double* C, double* Dm, double* G)
    {
        for (int i = 0; i < ns; i++) {
            double d[3], n2 = 0.0;
            for (int j = 0; j < 3; j++) { d[j] = P.obs_h * (x[j] - P.obs_c[i][j]); n2 += d[j] * d[j]; }
            const double nrm = sqrt(n2);
            s[i] = 1.0 - nrm;
            for (int j = 0; j < nx; j++) C[i * nx + j] = 0.0;
            for (int j = 0; j < 3; j++) C[i * nx + j] = -(P.obs_h * d[j]) / nrm;
            for (int j = 0; j < nu; j++) Dm[i * nu + j] = 0.0;
            G[i] = 0.0;
        }
    }
    SCP_DEV static void lin_rows(const Params&, double, int, double*, double*, double*) {}
    // (v_max, v), (w_max, w), (T_max, T), (M_max, M) in SOC (definition.jl:300-317, 354-371)
    SCP_DEV static void soc_rows(const Params& P, double, int, double* Mm, double* m)
    {
        constexpr int nz = nx + nu;
        for (int i = 0; i < nsoc * 4 * nz; i++) Mm[i] = 0.0;
        const int first[4] = {3, 10, nx + 0, nx + 3};
        const double bound[4] = {P.v_max, P.w_max, P.T_max, P.M_max};
        for (int c = 0; c < 4; c++) {
            for (int i = 0; i < 4; i++) m[4 * c + i] = 0.0;
            m[4 * c] = bound[c];
            for (int i = 0; i < 3; i++) Mm[(4 * c + 1 + i) * nz + first[c] + i] = 1.0;
        }
    }
    SCP_DEV static void glin_rows(const Params& P, double* Lp, double* lp)
    {
        Lp[0] = 1.0; lp[0] = -P.tf_max;
        Lp[1] = -1.0; lp[1] = P.tf_min;
    }
    SCP_DEV static void bc_ic(const Params&, const double* x, const double*, const double* pp, double* g, double* H, double* K)
    {
        for (int i = 0; i < nx; i++) { g[i] = x[i] - pp[i]; K[i] = 0.0; for (int j = 0; j < nx; j++) H[i * nx + j] = (i == j); }
    }
    SCP_DEV static void bc_tc(const Params&, const double* x, const double*, const double* pp, double* g, double* H, double* K)
    {
        for (int i = 0; i < nx; i++) { g[i] = x[i] - pp[nx + i]; K[i] = 0.0; for (int j = 0; j < nx; j++) H[i * nx + j] = (i == j); }
    }
    // Gamma = (1 - gamma)(T'T / T_max^2 + M'M / M_max^2), phi = gamma (tdil / tdil_max)^2 (+ the delta term) (:188-222)
    SCP_DEV static void cost_terms(const Params& P, double* Qu, double* lu, double* lx, double* tx, double* tp, double* Qp)
    {
        for (int i = 0; i < 3; i++) { Qu[i] = (1.0 - P.gamma) / (P.T_max * P.T_max); Qu[3 + i] = (1.0 - P.gamma) / (P.M_max * P.M_max); }
        for (int i = 0; i < nu; i++) lu[i] = 0.0;
        for (int i = 0; i < nx; i++) { lx[i] = 0.0; tx[i] = 0.0; }
        tp[0] = 0.0; Qp[0] = P.gamma / (P.tf_max * P.tf_max);
    }
};

}  // namespace scp
