// Quadrotor obstacle avoidance dynamics, test/examples/quadrotor/definition.jl:140-186:
// x=[r;v], u=[a;sigma], p=[tdil]; f=[v; a+g]*tdil, A[r,v]=I*tdil, B[v,a]=I*tdil,
// F[:,tdil] = f/tdil.  Gravity g=(0,0,-gnrm), parameters.jl:58-60,109.
#pragma once
#include "model_common.hpp"

namespace scp {

struct Quadrotor : ModelDefaults {
    static constexpr int id = 1;
    static constexpr int nx = 6, nu = 4, np = 1, npF = 1;
    // Jacobians A, B, F do not depend on (t, x, u) inside an interval -> variational discretize! kernel (K1v)
    static constexpr bool const_jacobian = true;
    static constexpr bool has_subproblem = true;   // false: discretize! / propagate / guess only (freeflyer.hpp)
    static constexpr bool structured = true;   // stage-structured PTR fast path available (stage_problem.hpp, ipm2_*.hpp)
    // largest normalised RK4 step 1/((N-1)(Nsub-1)) for which K1v matches the reference formulation to < 1e-10
    // (A is nilpotent: both RK4 forms are exact polynomials in h); coarser grids use the reference-form kernel K1
    static constexpr double var_form_max_step = 1e30;
    static constexpr bool s_input_free = true;   // s(t, k, x, p): admissible for GuSTO (gusto.jl:757-792)
    // the whole model is DATA (src/parser/problem.jl:64-121: traj.mdl is arbitrary user data) -- the constants of
    // test/examples/quadrotor/parameters.jl:96-130 cross the ABI in the blob:
    //   [gnrm, u_min, u_max, tilt_max (rad), tf_min, tf_max, gamma, obstacle 1: diag(H)(3), c(3), obstacle 2: diag(H)(3), c(3)]
    static constexpr int npar = 19;
    struct Params {
        double gnrm;
        double u_min, u_max, cos_tilt;
        double tf_min, tf_max, gamma;
        double obsH[2][3];  // diagonal of H
        double obsc[2][3];
    };
    static Params make_params(const double* par)
    {
        Params P;
        P.gnrm = par[0]; P.u_min = par[1]; P.u_max = par[2]; P.cos_tilt = cos(par[3]);
        P.tf_min = par[4]; P.tf_max = par[5]; P.gamma = par[6];
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) { P.obsH[i][j] = par[7 + 6 * i + j]; P.obsc[i][j] = par[10 + 6 * i + j]; }
        return P;
    }
    static constexpr int Fcol(int) { return 0; }

    SCP_DEV static void dyn(const Params& P, double, int, const double (&x)[nx], const double (&u)[nu],
                            const double* p, double (&f)[nx], double (&A)[nx * nx], double (&B)[nx * nu],
                            double (&Fc)[nx])
    {
        const double tdil = p[0];
        double f0[nx];
        f0[0] = x[3]; f0[1] = x[4]; f0[2] = x[5];
        f0[3] = u[0]; f0[4] = u[1]; f0[5] = u[2] - P.gnrm;
#pragma unroll
        for (int i = 0; i < nx; i++) f[i] = f0[i] * tdil;
        zero(A);
        A[0 + nx * 3] = tdil; A[1 + nx * 4] = tdil; A[2 + nx * 5] = tdil;
        zero(B);
        B[3 + nx * 0] = tdil; B[4 + nx * 1] = tdil; B[5 + nx * 2] = tdil;
#pragma unroll
        for (int i = 0; i < nx; i++) Fc[i] = f[i] / tdil;  // definition.jl:180
    }
    // structured products for the variational discretize! kernel: out = A v ; column j of B
    SCP_DEV static void Amul(const Params&, const double* p, const double (&v)[nx], double (&out)[nx])
    {
        const double tdil = p[0];
        out[0] = tdil * v[3]; out[1] = tdil * v[4]; out[2] = tdil * v[5]; out[3] = 0.0; out[4] = 0.0; out[5] = 0.0;
    }
    SCP_DEV static void Bcol(const Params&, const double* p, int j, double (&out)[nx])
    {
#pragma unroll
        for (int i = 0; i < nx; i++) out[i] = (j < 3 && i == 3 + j) ? p[0] : 0.0;
    }
    SCP_DEV static void action(double (&)[nx]) {}
    // IMPULSE discretisation (discretization.jl:186-193,384-390): impulsive velocity change dv = a (the first three
    // inputs), f(t, -k, x, u, p) = [0; a], B(t, -k, ...) = [0 0; I 0]; coasting (gravity only) between the nodes
    static constexpr bool has_fp32 = false;    // fp32 variant of K1 (scp_set_discretize_precision): Starship only
    static constexpr bool has_impulse = true;
    SCP_DEV static void impulse(const Params&, double, int, const double (&)[nx], const double (&u)[nu], const double*,
                                double (&dx)[nx], double (&B)[nx * nu])
    {
        dx[0] = 0.0; dx[1] = 0.0; dx[2] = 0.0; dx[3] = u[0]; dx[4] = u[1]; dx[5] = u[2];
        zero(B);
        B[3 + nx * 0] = 1.0; B[4 + nx * 1] = 1.0; B[5 + nx * 2] = 1.0;
    }
    // initial guess at node k of N (test/examples/quadrotor/definition.jl:60-90): straight-line state, hover input,
    // p = (tf_min + tf_max) / 2
    SCP_DEV static void guess(const Params& P, const double* pp, int N, int k, double (&x)[nx], double (&u)[nu], double* p, double*)
    {
        const double t = (double)k / (double)(N - 1), tg = (1.0 - t) * 0.0 + t * 1.0, c = (1.0 - tg) / (1.0 - 0.0);
#pragma unroll
        for (int i = 0; i < nx; i++) x[i] = c * pp[i] + (1.0 - c) * pp[nx + i];
        const double hov[4] = {0.0, 0.0, P.gnrm, P.gnrm};
#pragma unroll
        for (int i = 0; i < nu; i++) u[i] = c * hov[i] + (1.0 - c) * hov[i];
        p[0] = 0.5 * (P.tf_min + P.tf_max);
    }

    // ---- subproblem side: test/examples/quadrotor/definition.jl ----
    static constexpr int ns = 2, nl = 3, nsoc = 1, ng = 2, nic = 6, ntc = 6, npp = 12;  // pp = [r0 v0 rf vf]
    // s_i = 1 - ||H_i (r - c_i)||, C = -grad  (definition.jl:255-290, ellipsoid.jl:99-118)
    SCP_DEV static void s_eval(const Params& P, double, int, const double* x, const double*, const double*,
                               double* s, double* C, double* Dm, double* G)
    {
        for (int i = 0; i < ns; i++) {
            double d[3], n2 = 0.0;
            for (int j = 0; j < 3; j++) { d[j] = P.obsH[i][j] * (x[j] - P.obsc[i][j]); n2 += d[j] * d[j]; }
            const double nrm = fmax(sqrt(n2), 1e-300);   // the centre of an obstacle is not a differentiable point
            s[i] = 1.0 - nrm;
            for (int j = 0; j < nx; j++) C[i * nx + j] = 0.0;
            for (int j = 0; j < 3; j++) C[i * nx + j] = -(P.obsH[i][j] * d[j]) / nrm;  // -(H'H)(r-c)/||H(r-c)||
            for (int j = 0; j < nu; j++) Dm[i * nu + j] = 0.0;
            G[i] = 0.0;
        }
    }
    // U set (definition.jl:188-253): u_min <= sigma <= u_max, sigma cos(tilt) <= a3 ; (sigma, a) in SOC
    SCP_DEV static void lin_rows(const Params& P, double, int, double* L, double* Lp, double* l)
    {
        constexpr int nz = nx + nu;
        for (int i = 0; i < nl * nz; i++) L[i] = 0.0;
        for (int i = 0; i < nl; i++) Lp[i] = 0.0;
        L[0 * nz + nx + 3] = -1.0; l[0] = P.u_min;
        L[1 * nz + nx + 3] = 1.0; l[1] = -P.u_max;
        L[2 * nz + nx + 3] = P.cos_tilt; L[2 * nz + nx + 2] = -1.0; l[2] = 0.0;
    }
    SCP_DEV static void soc_rows(const Params&, double, int, double* Mm, double* m)
    {
        constexpr int nz = nx + nu;
        for (int i = 0; i < 4 * nz; i++) Mm[i] = 0.0;
        Mm[0 * nz + nx + 3] = 1.0; Mm[1 * nz + nx + 0] = 1.0; Mm[2 * nz + nx + 1] = 1.0; Mm[3 * nz + nx + 2] = 1.0;
        for (int i = 0; i < 4; i++) m[i] = 0.0;
    }
    // tdil - tf_max <= 0, tf_min - tdil <= 0 (the reference repeats them at every node; kept once)
    SCP_DEV static void glin_rows(const Params& P, double* Lp, double* lp)
    {
        Lp[0] = 1.0; lp[0] = -P.tf_max;
        Lp[1] = -1.0; lp[1] = P.tf_min;
    }
    SCP_DEV static void bc_ic(const Params&, const double* x, const double*, const double* pp, double* g, double* H,
                              double* K)
    {
        for (int i = 0; i < 6; i++) { g[i] = x[i] - pp[i]; K[i] = 0.0; for (int j = 0; j < 6; j++) H[i * 6 + j] = (i == j); }
    }
    SCP_DEV static void bc_tc(const Params&, const double* x, const double*, const double* pp, double* g, double* H,
                              double* K)
    {
        for (int i = 0; i < 6; i++) { g[i] = x[i] - pp[6 + i]; K[i] = 0.0; for (int j = 0; j < 6; j++) H[i * 6 + j] = (i == j); }
    }
    // Gamma = (1-gamma)(sigma/|g|)^2, phi = gamma (tdil/tdil_max)^2 (definition.jl:92-138)
    SCP_DEV static void cost_terms(const Params& P, double* Qu, double* lu, double* lx, double* tx, double* tp, double* Qp)
    {
        for (int i = 0; i < nu; i++) { Qu[i] = 0.0; lu[i] = 0.0; }
        Qu[3] = (1.0 - P.gamma) / (P.gnrm * P.gnrm);
        for (int i = 0; i < nx; i++) { lx[i] = 0.0; tx[i] = 0.0; }
        tp[0] = 0.0; Qp[0] = P.gamma / (P.tf_max * P.tf_max);
    }
};

}  // namespace scp
