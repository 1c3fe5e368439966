// 6-DoF free-flyer in the space station, test/examples/freeflyer/{parameters,definition}.jl:
// x = [r(3); v(3); q(4); w(3)] (quaternion q = [vector; scalar], src/utils/quaternion.jl:33-36), u = [T(3); M(3)],
// dynamics r' = v, v' = T/m, q' = 1/2 q (x) w, w' = J^-1 (M - w x J w), all times the time dilation p[0]
// (definition.jl:224-284), integration action q <- q/|q| after every RK4 step (:69-82) -- the first model with a real
// `action` and with a 13-dimensional state-dependent Jacobian.
//
// PARAMETERS.  p = [t_f; delta] with one room-SDF slack per room and node (np = 1 + 6 N, parameters.jl:121-128): ONE
// global parameter (np = 1) and np_node = 6 node parameters (model_common.hpp).  The slacks never enter the dynamics (F
// has the single structurally non-zero column of t_f); they appear in the six LINF room cones of X at their own node
// (definition.jl:334-346, lowered here to 6 x 6 linear rows like MOI's NormInfinity bridge), in the logsumexp row of s
// (:381-452) and in the terminal cost -eps_sdf sum(delta) (:172-184).
#pragma once
#include "model_common.hpp"

namespace scp {

struct Freeflyer : ModelDefaults {
    static constexpr int id = 4;
    static constexpr int nx = 13, nu = 6, np = 1, npF = 1;
    static constexpr int np_node = 6;                 // delta[:, k]: one room-SDF slack per room of the station
    static constexpr int n_obs = 3, n_iss = 6;        // ellipsoidal obstacles / rooms (parameters.jl:95-109)
    static constexpr bool const_jacobian = false;
    static constexpr double var_form_max_step = 0.0;
    static constexpr bool structured = false;
    // variational form of discretize! (K1x) for physical RK4 steps up to 0.047 s: measured against the reference formulation on
    // strongly perturbed trajectories at N = 200, Nsub = 15 -- B-, B+ 4.3e-11 (t_f = 110 s), 7.1e-11 (130 s = 0.0467 s steps),
    // 1.1e-10 (150 s), 2.6e-10 (200 s); every other block below 1e-12.  Coarser steps use the reference formulation (K1).
    static constexpr double var_form_max_phys_step = 0.047;
    static constexpr bool has_subproblem = true;
    static constexpr bool global_rows_in_X = true;    // t_f bounds are members of X (definition.jl:318-331)
    static constexpr int linf_groups = n_iss, linf_rows = 6;
    static constexpr bool s_input_free = true;
    // the whole model is DATA (src/parser/problem.jl:64-121: traj.mdl is arbitrary user data): vehicle, trajectory and
    // environment constants of parameters.jl:86-192, laid out as
    //   [m, J(3), v_max, w_max, T_max, M_max, tf_min, tf_max, gamma, hom, eps_sdf, obstacles n_obs x (h, c(3)),
    //    rooms n_iss x (c(3), s(3))]
    static constexpr int npar = 13 + 4 * n_obs + 6 * n_iss;

    struct Params {
        double m, J[3];
        double v_max, w_max, T_max, M_max;       // parameters.jl:135-138
        double tf_min, tf_max, gamma, hom, eps_sdf;   // parameters.jl:168-172
        double obs_h[n_obs], obs_c[n_obs][3];    // Ellipsoid(H = h I, c), parameters.jl:95-101
        double room_c[n_iss][3], room_s[n_iss][3];    // Hyperrectangle scaling x = s .* y + c, hyperrectangle.jl:26-54
    };
    static Params make_params(const double* par)
    {
        Params P;
        P.m = par[0]; P.J[0] = par[1]; P.J[1] = par[2]; P.J[2] = par[3];
        P.v_max = par[4]; P.w_max = par[5]; P.T_max = par[6]; P.M_max = par[7];
        P.tf_min = par[8]; P.tf_max = par[9]; P.gamma = par[10]; P.hom = par[11]; P.eps_sdf = par[12];
        const double* o = par + 13;
        for (int i = 0; i < n_obs; i++) { P.obs_h[i] = o[4 * i]; for (int j = 0; j < 3; j++) P.obs_c[i][j] = o[4 * i + 1 + j]; }
        const double* r = o + 4 * n_obs;
        for (int i = 0; i < n_iss; i++) for (int j = 0; j < 3; j++) { P.room_c[i][j] = r[6 * i + j]; P.room_s[i][j] = r[6 * i + 3 + j]; }
        return P;
    }
    static constexpr int Fcol(int) { return 0; }
    template <class PP>
    SCP_DEV static double time_dilation(const PP&, const double* p) { return p[0]; }

    SCP_DEV static void cross(const double* a, const double* b, double* c)
    {
        c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
    }
    // translation (r, v) and attitude (q, w) do not see each other (:224-284): A = blockdiag(A_rv, A_qw), A_rv = [0, tdil I; 0, 0],
    // so Phi = blockdiag([I, a I; 0, I], Phi_qw): only the 7 x 7 attitude block needs the elimination
    static constexpr int lu_lead = 6;
    static constexpr bool lu_decoupled = true;
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
    // out = A(x, p) v without forming the 13 x 13 matrix (39 structural non-zeros, the same entries as dyn() writes): the
    // variational discretize! kernel (K1x) multiplies A by one column per RK4 stage
    static constexpr bool has_amulx = true;
    SCP_DEV static void Amulx(const Params& P, const double* p, const double (&x)[nx], const double (&v)[nx], double (&out)[nx])
    {
        const double td = p[0];
        const double qv[3] = {x[6], x[7], x[8]}, qw = x[9], w[3] = {x[10], x[11], x[12]};
        const double a[3] = {v[6], v[7], v[8]}, a4 = v[9], b[3] = {v[10], v[11], v[12]};
        double wxa[3], qxb[3], Jb[3], wxJb[3], Jw[3], Jwxb[3];
        cross(w, a, wxa); cross(qv, b, qxb);
#pragma unroll
        for (int i = 0; i < 3; i++) { Jb[i] = P.J[i] * b[i]; Jw[i] = P.J[i] * w[i]; }
        cross(w, Jb, wxJb); cross(Jw, b, Jwxb);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            out[i] = td * v[3 + i];
            out[3 + i] = 0.0;
            // rows 6..8: -1/2 [w]x a + 1/2 w a4 + 1/2 (qw b + [qv]x b)
            out[6 + i] = td * (-0.5 * wxa[i] + 0.5 * w[i] * a4 + 0.5 * (qw * b[i] + qxb[i]));
            // rows 10..12: -J^-1 ([w]x J b - [J w]x b)
            out[10 + i] = -td * (wxJb[i] - Jwxb[i]) / P.J[i];
        }
        out[9] = -0.5 * td * (w[0] * a[0] + w[1] * a[1] + w[2] * a[2] + qv[0] * b[0] + qv[1] * b[1] + qv[2] * b[2]);
    }
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

    // per-problem data pp = [r0 v0 q0 w0 rf vf qf wf] (npp = 26, below)

    // q0' (x) q1 with q = [v; w]; Log(q) -> (angle, axis)  (quaternion.jl:211-214, 257-260, 277-282)
    SCP_DEV static void qmul(const double* a, const double* b, double* r)
    {
        double axb[3];
        cross(a, b, axb);
        for (int i = 0; i < 3; i++) r[i] = a[3] * b[i] + b[3] * a[i] + axb[i];
        r[3] = a[3] * b[3] - (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
    }
    // initial guess at node k of N (definition.jl:84-186): an axis-by-axis (L1) path at constant speed, SLERP attitude,
    // constant body rate, idle inputs, p = (tf_min + tf_max) / 2; pn = the node's room slacks delta[i, k] = 1 - |(r_k - c_i) ./
    // s_i|_inf, i.e. the signed distances of the guess (:166-172)
    SCP_DEV static void guess(const Params& P, const double* pp, int N, int k, double (&x)[nx], double (&u)[nu], double* p,
                              double* pn)
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
        p[0] = T;
        for (int i = 0; i < n_iss; i++) {
            double d = 0.0;
            for (int j = 0; j < 3; j++) d = fmax(d, fabs((x[j] - P.room_c[i][j]) / P.room_s[i][j]));
            pn[i] = 1.0 - d;
        }
    }

    // ---- subproblem side: test/examples/freeflyer/definition.jl ----
    static constexpr int ns = n_obs + 1, nl = n_iss * 6, nsoc = 4, ng = 2, nic = 13, ntc = 13, npp = 26;
    // s = [1 - ||H_i (r - c_i)|| (obstacles); -logsumexp_hom(delta[:, k])]  (definition.jl:381-452, ellipsoid.jl:99-118,
    // helper.jl:623-651).  G is COMPACT: ns x (np + np_node), column 1 + i = d/d delta[i, k].
    SCP_DEV static void s_eval(const Params& P, double, int k, const double* x, const double*, const double* p, double* s,
                               double* C, double* Dm, double* G)
    {
        constexpr int npc = np + np_node;
        for (int i = 0; i < ns * nx; i++) C[i] = 0.0;
        for (int i = 0; i < ns * nu; i++) Dm[i] = 0.0;
        for (int i = 0; i < ns * npc; i++) G[i] = 0.0;
        for (int i = 0; i < n_obs; i++) {
            double d[3], n2 = 0.0;
            for (int j = 0; j < 3; j++) { d[j] = P.obs_h[i] * (x[j] - P.obs_c[i][j]); n2 += d[j] * d[j]; }
            const double nrm = fmax(sqrt(n2), 1e-300);       // the centre of an obstacle is not a differentiable point
            s[i] = 1.0 - nrm;
            for (int j = 0; j < 3; j++) C[i * nx + j] = -(P.obs_h[i] * d[j]) / nrm;
        }
        const double* dl = p + np + (long)np_node * (k - 1);
        double a = -1e300;
        for (int i = 0; i < n_iss; i++) a = fmax(a, P.hom * dl[i]);
        double E = 0.0, e[n_iss];
        for (int i = 0; i < n_iss; i++) { e[i] = exp(P.hom * dl[i] - a); E += e[i]; }
        s[n_obs] = -(a + log(E)) / P.hom;
        for (int i = 0; i < n_iss; i++) G[n_obs * npc + np + i] = -e[i] / E;
    }
    // X: the six LINF room cones (1 - delta_ik, (r - c_i) ./ s_i) lowered to rows (definition.jl:334-346; MOI NormInfinity
    // bridge: +-y_j - t <= 0, the three "+" rows of a room first).  Lp is COMPACT (nl x (np + np_node)).
    SCP_DEV static void lin_rows(const Params& P, double, int, double* L, double* Lp, double* l)
    {
        constexpr int nz = nx + nu, npc = np + np_node;
        for (int i = 0; i < nl * nz; i++) L[i] = 0.0;
        for (int i = 0; i < nl * npc; i++) Lp[i] = 0.0;
        for (int i = 0; i < n_iss; i++)
            for (int sgn = 0; sgn < 2; sgn++)
                for (int j = 0; j < 3; j++) {
                    const int row = 6 * i + 3 * sgn + j;
                    const double sg = sgn == 0 ? 1.0 : -1.0;
                    L[row * nz + j] = sg / P.room_s[i][j];
                    Lp[row * npc + np + i] = 1.0;                      // -(1 - delta_ik) = delta_ik - 1
                    l[row] = -sg * P.room_c[i][j] / P.room_s[i][j] - 1.0;
                }
    }
    // (v_max, v), (w_max, w) in X; (T_max, T), (M_max, M) in U (definition.jl:300-317, 354-371)
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
    // Gamma = (1 - gamma)(T'T / T_max^2 + M'M / M_max^2), phi = gamma (tdil / tdil_max)^2 - eps_sdf sum(delta) (:172-222).
    // tp, Qp are COMPACT (np + np_node): the node-parameter entries apply to the parameters of EVERY node.
    SCP_DEV static void cost_terms(const Params& P, double* Qu, double* lu, double* lx, double* tx, double* tp, double* Qp)
    {
        for (int i = 0; i < 3; i++) { Qu[i] = (1.0 - P.gamma) / (P.T_max * P.T_max); Qu[3 + i] = (1.0 - P.gamma) / (P.M_max * P.M_max); }
        for (int i = 0; i < nu; i++) lu[i] = 0.0;
        for (int i = 0; i < nx; i++) { lx[i] = 0.0; tx[i] = 0.0; }
        tp[0] = 0.0; Qp[0] = P.gamma / (P.tf_max * P.tf_max);
        for (int i = 0; i < np_node; i++) { tp[np + i] = -P.eps_sdf; Qp[np + i] = 0.0; }
    }
};

}  // namespace scp
