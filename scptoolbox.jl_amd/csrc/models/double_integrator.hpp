// Double integrator with friction as a TrajectoryProblem (builder-defined: the
// reference solves it by LCvx only).  Physical model from
// test/examples/double_integrator/parameters.jl:58-64: f = [x2; u - g], fixed
// duration T, so in normalised time xdot = T*[x2; u - g].  np = 0.
#pragma once
#include "model_common.hpp"

namespace scp {

struct DoubleIntegrator : ModelDefaults {
    static constexpr int id = 0;
    static constexpr int nx = 2, nu = 1, np = 0, npF = 0;
    // Jacobians A, B, F do not depend on (t, x, u) inside an interval -> variational discretize! kernel (K1v)
    static constexpr bool const_jacobian = true;
    static constexpr bool has_subproblem = true;   // false: discretize! / propagate / guess only (freeflyer.hpp)
    static constexpr bool structured = true;   // stage-structured PTR fast path available (stage_problem.hpp, ipm2_*.hpp)
    // largest normalised RK4 step 1/((N-1)(Nsub-1)) for which K1v matches the reference formulation to < 1e-10
    // (A is nilpotent: both RK4 forms are exact polynomials in h); coarser grids use the reference-form kernel K1
    static constexpr double var_form_max_step = 1e30;
    static constexpr int npar = 2;  // [g, T]
    struct Params {
        double g, T;
    };
    static Params make_params(const double* par) { return Params{par[0], par[1]}; }
    static constexpr int Fcol(int) { return 0; }

    // f, A (col-major nx*nx), B (nx*nu), Fc (nx*npF)
    SCP_DEV static void dyn(const Params& P, double, int, const double (&x)[nx], const double (&u)[nu],
                            const double*, double (&f)[nx], double (&A)[nx * nx], double (&B)[nx * nu],
                            double (&Fc)[nx])
    {
        f[0] = P.T * x[1];
        f[1] = P.T * (u[0] - P.g);
        A[0] = 0.0; A[1] = 0.0; A[2] = P.T; A[3] = 0.0;
        B[0] = 0.0; B[1] = P.T;
        Fc[0] = 0.0; Fc[1] = 0.0;
    }
    // structured products for the variational discretize! kernel: out = A v ; column j of B
    SCP_DEV static void Amul(const Params& P, const double*, const double (&v)[nx], double (&out)[nx]) { out[0] = P.T * v[1]; out[1] = 0.0; }
    SCP_DEV static void Bcol(const Params& P, const double*, int, double (&out)[nx]) { out[0] = 0.0; out[1] = P.T; }
    SCP_DEV static void action(double (&)[nx]) {}
    // IMPULSE discretisation (src/solvers/discretization.jl:186-193,384-390: the model is evaluated with k < 0): the
    // input is an impulsive velocity change, f(t, -k, x, u, p) = [0; u], B(t, -k, ...) = [0; 1]; between the nodes the
    // system coasts (u = 0).  Same convention as the reference's oscillator example (oscillator/definition.jl:170-186).
    static constexpr bool has_fp32 = false;    // fp32 variant of K1 (scp_set_discretize_precision): Starship only
    static constexpr bool has_impulse = true;
    SCP_DEV static void impulse(const Params&, double, int, const double (&)[nx], const double (&u)[nu], const double*,
                                double (&dx)[nx], double (&B)[nx * nu])
    {
        dx[0] = 0.0; dx[1] = u[0];
        B[0] = 0.0; B[1] = 1.0;
    }
    // initial guess at node k (0-based) of N, traj.guess (problem.jl:686-700): straight line between the boundary
    // states (helper.jl:203-219), accelerate-then-brake input (a one-signed |u| >= 1 guess can never brake)
    SCP_DEV static void guess(const Params&, const double* pp, int N, int k, double (&x)[nx], double (&u)[nu], double*, double*)
    {
        const double t = (double)k / (double)(N - 1), tg = (1.0 - t) * 0.0 + t * 1.0, c = (1.0 - tg) / (1.0 - 0.0);
        x[0] = c * pp[0] + (1.0 - c) * pp[2]; x[1] = c * pp[1] + (1.0 - c) * pp[3];
        u[0] = k < N / 2 ? 1.5 : -1.5;
    }

    // ---- subproblem side (builder-defined PTR problem): |u| <= 2 convex, 1 - u^2 <= 0 in s,
    //      cost int u^2, x(0) = pp[0:2], x(1) = pp[2:4] ----
    static constexpr int ns = 1, nl = 2, nsoc = 0, ng = 0, nic = 2, ntc = 2, npp = 4;
    SCP_DEV static void s_eval(const Params&, double, int, const double* x, const double* u, const double*,
                               double* s, double* C, double* Dm, double* G)
    {
        (void)x; (void)G;
        s[0] = 1.0 - u[0] * u[0];
        C[0] = 0.0; C[1] = 0.0;
        Dm[0] = -2.0 * u[0];
    }
    // lin rows: L z + Lp p + l <= 0, z = (x, u) physical; row-major L[nl][nz]
    SCP_DEV static void lin_rows(const Params&, double, int, double* L, double* Lp, double* l)
    {
        (void)Lp;
        L[0] = 0; L[1] = 0; L[2] = 1.0; l[0] = -2.0;   //  u - 2 <= 0
        L[3] = 0; L[4] = 0; L[5] = -1.0; l[1] = -2.0;  // -u - 2 <= 0
    }
    SCP_DEV static void soc_rows(const Params&, double, int, double*, double*) {}
    SCP_DEV static void glin_rows(const Params&, double*, double*) {}
    SCP_DEV static void bc_ic(const Params&, const double* x, const double*, const double* pp, double* g, double* H,
                              double* K)
    {
        (void)K;
        g[0] = x[0] - pp[0]; g[1] = x[1] - pp[1];
        H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 1;
    }
    SCP_DEV static void bc_tc(const Params&, const double* x, const double*, const double* pp, double* g, double* H,
                              double* K)
    {
        (void)K;
        g[0] = x[0] - pp[2]; g[1] = x[1] - pp[3];
        H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 1;
    }
    // cost: Gamma = sum Qu u^2 + lu'u + lx'x ; phi = tx'x_N + tp'p + sum Qp p^2
    SCP_DEV static void cost_terms(const Params&, double* Qu, double* lu, double* lx, double* tx, double* tp, double* Qp)
    {
        (void)tp; (void)Qp;
        Qu[0] = 1.0; lu[0] = 0.0; lx[0] = 0; lx[1] = 0; tx[0] = 0; tx[1] = 0;
    }
};

}  // namespace scp
