/*
 * scp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's `discretize!` hot path
 * (UW-ACL/SCPToolbox.jl v1.0.0).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or call this file.  The product path
 * (scptoolbox.jl_amd/csrc) never includes or links it.
 *
 * PARITY STATUS: "parity unpinned" -- the reference ships no golden vectors
 * for this path and Julia is absent, so the reference cannot
 * be executed here.  This restatement is pinned instead on mathematics
 * (tests/test_oracle_discretize.py): LTI closed forms (matrix exponential),
 * finite-difference Jacobians, and the reference's own second, independent
 * FOH discretiser for the double integrator
 * (test/examples/double_integrator/parameters.jl:64-78).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Arrays are column-major exactly like Julia's.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_MAX_NX 16

typedef struct {
    int nx, nu, np;
    /* (t,k,x,u,p,par) -> out ; matrices column-major */
    void (*f)(double, int, const double *, const double *, const double *, const double *, double *);
    void (*A)(double, int, const double *, const double *, const double *, const double *, double *);
    void (*B)(double, int, const double *, const double *, const double *, const double *, double *);
    void (*F)(double, int, const double *, const double *, const double *, const double *, double *);
    /* post-step integration action on the state part (helper.jl:494-496); may be NULL */
    void (*action)(double *x);
} oracle_model;

/* ------------------------------------------------------------------------ */
/* Models                                                                    */
/* ------------------------------------------------------------------------ */

/* Double integrator with friction: test/examples/double_integrator/parameters.jl:58-64
 * f = (t,x,u) -> [x[2]; u - g]; as a TrajectoryProblem (builder-defined)
 * the normalised-time dynamics are T*f with fixed duration T (np = 0).
 * par = [g, T]. */
static void di_f(double t, int k, const double *x, const double *u, const double *p, const double *par, double *f)
{
    (void)t; (void)p;
    if (k < 0) { f[0] = 0.0; f[1] = u[0]; return; }   /* impulse evaluation (discretization.jl:191): dv = u */
    f[0] = par[1] * x[1];
    f[1] = par[1] * (u[0] - par[0]);
}
static void di_A(double t, int k, const double *x, const double *u, const double *p, const double *par, double *A)
{
    (void)t; (void)k; (void)x; (void)u; (void)p;
    A[0] = 0; A[1] = 0; A[2] = par[1]; A[3] = 0; /* A[0,1] = T */
}
static void di_B(double t, int k, const double *x, const double *u, const double *p, const double *par, double *B)
{
    (void)t; (void)x; (void)u; (void)p;
    B[0] = 0; B[1] = k < 0 ? 1.0 : par[1];            /* k < 0: impulse input matrix (discretization.jl:389) */
}
static void di_F(double t, int k, const double *x, const double *u, const double *p, const double *par, double *F)
{
    (void)t; (void)k; (void)x; (void)u; (void)p; (void)par; (void)F; /* np = 0 */
}

/* Quadrotor: test/examples/quadrotor/definition.jl:140-186.
 * x=[r;v], u=[a(3);sigma], p=[tdil]; f=[v; a+g]*tdil; A[r,v]=I*tdil; B[v,a]=I*tdil;
 * F[:,1] = f/tdil.  par = [gnrm]  (g = (0,0,-gnrm), parameters.jl:58-60). */
static void quad_f(double t, int k, const double *x, const double *u, const double *p, const double *par, double *f)
{
    (void)t;
    double tdil = p[0];
    if (k < 0) { f[0] = f[1] = f[2] = 0.0; f[3] = u[0]; f[4] = u[1]; f[5] = u[2]; return; }   /* impulse: dv = a */
    f[0] = x[3]; f[1] = x[4]; f[2] = x[5];
    f[3] = u[0] + 0.0; f[4] = u[1] + 0.0; f[5] = u[2] + (-par[0]);
    for (int i = 0; i < 6; i++) f[i] *= tdil;
}
static void quad_A(double t, int k, const double *x, const double *u, const double *p, const double *par, double *A)
{
    (void)t; (void)k; (void)x; (void)u; (void)par;
    memset(A, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++) A[i + 6 * (3 + i)] = 1.0 * p[0];
}
static void quad_B(double t, int k, const double *x, const double *u, const double *p, const double *par, double *B)
{
    (void)t; (void)x; (void)u; (void)par;
    memset(B, 0, 24 * sizeof(double));
    for (int i = 0; i < 3; i++) B[(3 + i) + 6 * i] = k < 0 ? 1.0 : 1.0 * p[0];
}
static void quad_F(double t, int k, const double *x, const double *u, const double *p, const double *par, double *F)
{
    double f[6];
    quad_f(t, k, x, u, p, par, f);
    for (int i = 0; i < 6; i++) F[i] = f[i] / p[0]; /* definition.jl:180 */
}

/* Rocket landing (builder-defined TrajectoryProblem over the physical model of
 * test/examples/rocket_landing/parameters.jl:77-146): x=[r;v;z], u=[a(3);xi], p=[tf];
 * xdot = tf*(A_c x + B_c u + p_c), A_c = [0 I 0; -(w^x)^2 -2 w^x 0; 0],
 * B_c = [0; I 0; 0 -alpha], p_c = [0; g; 0]  (parameters.jl:108-120).
 * par = [g(3), omega(3), alpha]. */
static void rocket_Ac(const double *par, double *Ac /* 7x7 col-major */)
{
    const double *w = par + 3;
    double S[9] = {0, w[2], -w[1], -w[2], 0, w[0], w[1], -w[0], 0}; /* skew(w), col-major */
    double S2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double a = 0;
            for (int l = 0; l < 3; l++) a += S[i + 3 * l] * S[l + 3 * j];
            S2[i + 3 * j] = a;
        }
    memset(Ac, 0, 49 * sizeof(double));
    for (int i = 0; i < 3; i++) Ac[i + 7 * (3 + i)] = 1.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            Ac[(3 + i) + 7 * j] = -S2[i + 3 * j];
            Ac[(3 + i) + 7 * (3 + j)] = -2.0 * S[i + 3 * j];
        }
}
static void rocket_f(double t, int k, const double *x, const double *u, const double *p, const double *par, double *f)
{
    (void)t; (void)k;
    double Ac[49];
    rocket_Ac(par, Ac);
    for (int i = 0; i < 7; i++) {
        double a = 0;
        for (int j = 0; j < 7; j++) a += Ac[i + 7 * j] * x[j];
        f[i] = a;
    }
    for (int i = 0; i < 3; i++) f[3 + i] += u[i] + par[i];
    f[6] += -par[6] * u[3];
    for (int i = 0; i < 7; i++) f[i] *= p[0];
}
static void rocket_A(double t, int k, const double *x, const double *u, const double *p, const double *par, double *A)
{
    (void)t; (void)k; (void)x; (void)u;
    rocket_Ac(par, A);
    for (int i = 0; i < 49; i++) A[i] *= p[0];
}
static void rocket_B(double t, int k, const double *x, const double *u, const double *p, const double *par, double *B)
{
    (void)t; (void)k; (void)x; (void)u;
    memset(B, 0, 28 * sizeof(double));
    for (int i = 0; i < 3; i++) B[(3 + i) + 7 * i] = p[0];
    B[6 + 7 * 3] = -par[6] * p[0];
}
static void rocket_F(double t, int k, const double *x, const double *u, const double *p, const double *par, double *F)
{
    double f[7];
    rocket_f(t, k, x, u, p, par, f);
    for (int i = 0; i < 7; i++) F[i] = f[i] / p[0];
}

/* Starship landing flip: test/examples/starship_flip/definition.jl:498-637 (dynamics + Jacobians),
 * parameters.jl:99-212.  x=[r(2);v(2);theta;omega;m;delta_d], u=[T;delta;delta_dot], p=[t1;t2;xs(8)].
 * par = [N] (unused by the dynamics). */
#define SS_G0 9.81
#define SS_M 120e3
#define SS_LCG (0.4 * 50.0)
#define SS_LCP (0.45 * 50.0)
#define SS_J (1.0 / 12.0 * SS_M * (6.0 * 4.5 * 4.5 + 50.0 * 50.0))
#define SS_CD (SS_M * SS_G0 / (85.0 * 85.0) * 1.2)
#define SS_ALPHA_E (-1.0 / (330.0 * SS_G0))
#define SS_RATE_DELAY 0.05
#define SS_TAU_S 0.5
static double ss_tdil(double t, const double *p) { return (t <= SS_TAU_S) ? p[0] / SS_TAU_S : p[1] / (1.0 - SS_TAU_S); }
static void ss_f(double t, int k, const double *x, const double *u, const double *p, const double *par, double *f)
{
    (void)k; (void)par;
    const double *v = x + 2;
    double th = x[4], om = x[5], dd = x[7], T = u[0], de = u[1];
    double leng = -SS_LCG, lcp = SS_LCP - SS_LCG;
    double ei[2] = {cos(th), sin(th)}, ej[2] = {-sin(th), cos(th)};
    double Tv[2], D[2], nv = sqrt(v[0] * v[0] + v[1] * v[1]);
    for (int i = 0; i < 2; i++) { Tv[i] = T * (-sin(de) * ei[i] + cos(de) * ej[i]); D[i] = -SS_CD * nv * v[i]; }
    double MT = leng * T * sin(de);
    double MD = -lcp * (D[0] * ei[0] + D[1] * ei[1]);
    f[0] = v[0]; f[1] = v[1];
    f[2] = (Tv[0] + D[0]) / SS_M + 0.0;
    f[3] = (Tv[1] + D[1]) / SS_M - SS_G0;
    f[4] = om;
    f[5] = (MT + MD) / SS_J;
    f[6] = SS_ALPHA_E * T;
    f[7] = (de - dd) / SS_RATE_DELAY;
    double tdil = ss_tdil(t, p);
    for (int i = 0; i < 8; i++) f[i] *= tdil;
}
static void ss_A(double t, int k, const double *x, const double *u, const double *p, const double *par, double *A)
{
    (void)k; (void)par;
    const double *v = x + 2;
    double th = x[4], T = u[0], de = u[1];
    double lcp = SS_LCP - SS_LCG;
    double ei[2] = {cos(th), sin(th)}, ej[2] = {-sin(th), cos(th)};
    double nv = sqrt(v[0] * v[0] + v[1] * v[1]);
    double D[2] = {-SS_CD * nv * v[0], -SS_CD * nv * v[1]};
    double gD[2][2];                                   /* grad_v D = -CD (|v| I + v v'/|v|)  (:572) */
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) gD[i][j] = -SS_CD * ((i == j ? nv : 0.0) + v[i] * v[j] / nv);
    double gthTv[2] = {T * (-sin(de) * ej[0] + cos(de) * -ei[0]), T * (-sin(de) * ej[1] + cos(de) * -ei[1])};
    double gvMD[2];                                    /* -lcp * grad_v D' * ei  (:573) */
    for (int j = 0; j < 2; j++) gvMD[j] = -lcp * (gD[0][j] * ei[0] + gD[1][j] * ei[1]);
    double gthMD = -lcp * (D[0] * ej[0] + D[1] * ej[1]);
    memset(A, 0, 64 * sizeof(double));
    A[0 + 8 * 2] = 1.0; A[1 + 8 * 3] = 1.0;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) A[(2 + i) + 8 * (2 + j)] = gD[i][j] / SS_M;
    A[2 + 8 * 4] = gthTv[0] / SS_M; A[3 + 8 * 4] = gthTv[1] / SS_M;
    A[4 + 8 * 5] = 1.0;
    A[5 + 8 * 2] = gvMD[0] / SS_J; A[5 + 8 * 3] = gvMD[1] / SS_J;
    A[5 + 8 * 4] = gthMD / SS_J;
    A[7 + 8 * 7] = -1.0 / SS_RATE_DELAY;
    double tdil = ss_tdil(t, p);
    for (int i = 0; i < 64; i++) A[i] *= tdil;
}
static void ss_B(double t, int k, const double *x, const double *u, const double *p, const double *par, double *B)
{
    (void)k; (void)par;
    double th = x[4], T = u[0], de = u[1];
    double leng = -SS_LCG;
    double ei[2] = {cos(th), sin(th)}, ej[2] = {-sin(th), cos(th)};
    memset(B, 0, 24 * sizeof(double));
    for (int i = 0; i < 2; i++) {
        B[(2 + i) + 8 * 0] = (-sin(de) * ei[i] + cos(de) * ej[i]) / SS_M;
        B[(2 + i) + 8 * 1] = T * (-cos(de) * ei[i] - sin(de) * ej[i]) / SS_M;
    }
    B[5 + 8 * 0] = leng * sin(de) / SS_J;
    B[5 + 8 * 1] = leng * T * cos(de) / SS_J;
    B[6 + 8 * 0] = SS_ALPHA_E;
    B[7 + 8 * 1] = 1.0 / SS_RATE_DELAY;
    double tdil = ss_tdil(t, p);
    for (int i = 0; i < 24; i++) B[i] *= tdil;
}
static void ss_F(double t, int k, const double *x, const double *u, const double *p, const double *par, double *F)
{
    double f[8];
    int id_t = (t <= SS_TAU_S) ? 0 : 1;
    ss_f(t, k, x, u, p, par, f);
    memset(F, 0, 80 * sizeof(double));
    for (int i = 0; i < 8; i++) F[i + 8 * id_t] = f[i] / p[id_t];   /* :632 */
}

/* ------------------------------------------------------------------------ */
/* 6-DoF free-flyer: test/examples/freeflyer/definition.jl:224-284 (dynamics and Jacobians), parameters.jl:140-141
 * (m, J), quaternion algebra of src/utils/quaternion.jl (q = [v; w], scalar LAST; skew :190-198, product :211-214).
 * x = [r(3); v(3); q(4); w(3)], u = [T(3); M(3)], p = [t_f] -- the reference's parameter vector also carries the
 * room-SDF slacks delta (np = 1 + 6N), which never enter the dynamics: F has the single structurally non-zero
 * column of t_f, so discretize! is restated with np = 1.  Integration action: q <- q / |q| after every full RK4 step
 * (definition.jl:69-82).  par = [m, J1, J2, J3]. */
static void ff_cross(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
static void ff_f(double t, int k, const double *x, const double *u, const double *p, const double *par, double *f)
{
    (void)t; (void)k;
    const double *v = x + 3, *q = x + 6, *w = x + 10, *T = u, *M = u + 3;
    double m = par[0], J[3] = {par[1], par[2], par[3]};
    double qv_x_w[3];
    ff_cross(q, w, qv_x_w);
    for (int i = 0; i < 3; i++) { f[i] = v[i]; f[3 + i] = T[i] / m; }
    /* 0.5 vec(q * w): skew(q, :L) [w; 0] = [q.w w + q.v x w; -q.v . w]  (:238) */
    for (int i = 0; i < 3; i++) f[6 + i] = 0.5 * (q[3] * w[i] + qv_x_w[i]);
    f[9] = -0.5 * (q[0] * w[0] + q[1] * w[1] + q[2] * w[2]);
    double Jw[3] = {J[0] * w[0], J[1] * w[1], J[2] * w[2]}, wxJw[3];
    ff_cross(w, Jw, wxJw);
    for (int i = 0; i < 3; i++) f[10 + i] = (M[i] - wxJw[i]) / J[i];      /* J \ (M - w x J w)  (:239) */
    for (int i = 0; i < 13; i++) f[i] *= p[0];
}
static void ff_skew3(const double *a, double *S) /* column-major 3x3, helper.jl:65-70 */
{
    for (int i = 0; i < 9; i++) S[i] = 0.0;
    S[0 + 3 * 1] = -a[2]; S[0 + 3 * 2] = a[1]; S[1 + 3 * 2] = -a[0];
    S[1 + 3 * 0] = a[2]; S[2 + 3 * 0] = -a[1]; S[2 + 3 * 1] = a[0];
}
static void ff_A(double t, int k, const double *x, const double *u, const double *p, const double *par, double *A)
{
    (void)t; (void)k; (void)u;
    const double *q = x + 6, *w = x + 10;
    double J[3] = {par[1], par[2], par[3]};
    memset(A, 0, 169 * sizeof(double));
    for (int i = 0; i < 3; i++) A[i + 13 * (3 + i)] = 1.0;
    /* dfq/dq = 0.5 skew(Quaternion(w), :R) = 0.5 [[-[w]x, w]; [-w', 0]]  (:250) */
    double Sw[9];
    ff_skew3(w, Sw);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) A[(6 + i) + 13 * (6 + j)] = -0.5 * Sw[i + 3 * j];
        A[(6 + i) + 13 * 9] = 0.5 * w[i];
        A[9 + 13 * (6 + i)] = -0.5 * w[i];
    }
    /* dfq/dw = 0.5 skew(q)[:, 1:3] = 0.5 [q.w I + [q.v]x; -q.v']  (:251) */
    double Sq[9];
    ff_skew3(q, Sq);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) A[(6 + i) + 13 * (10 + j)] = 0.5 * ((i == j ? q[3] : 0.0) + Sq[i + 3 * j]);
        A[9 + 13 * (10 + i)] = -0.5 * q[i];
    }
    /* dfw/dw = -J \ (skew(w) J - skew(J w))  (:252) */
    double Jw[3] = {J[0] * w[0], J[1] * w[1], J[2] * w[2]}, SJw[9];
    ff_skew3(Jw, SJw);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) A[(10 + i) + 13 * (10 + j)] = -(Sw[i + 3 * j] * J[j] - SJw[i + 3 * j]) / J[i];
    for (int i = 0; i < 169; i++) A[i] *= p[0];
}
static void ff_B(double t, int k, const double *x, const double *u, const double *p, const double *par, double *B)
{
    (void)t; (void)k; (void)x; (void)u;
    memset(B, 0, 78 * sizeof(double));
    for (int i = 0; i < 3; i++) { B[(3 + i) + 13 * i] = 1.0 / par[0]; B[(10 + i) + 13 * (3 + i)] = 1.0 / par[1 + i]; }
    for (int i = 0; i < 78; i++) B[i] *= p[0];
}
static void ff_F(double t, int k, const double *x, const double *u, const double *p, const double *par, double *F)
{
    ff_f(t, k, x, u, p, par, F);
    for (int i = 0; i < 13; i++) F[i] /= p[0];      /* F[:, id_t] = f / tdil  (:278) */
}
static void ff_action(double *x)
{
    double n = sqrt(x[6] * x[6] + x[7] * x[7] + x[8] * x[8] + x[9] * x[9]);
    for (int i = 6; i < 10; i++) x[i] /= n;
}

static const oracle_model MODELS[] = {
    {2, 1, 0, di_f, di_A, di_B, di_F, NULL},
    {6, 4, 1, quad_f, quad_A, quad_B, quad_F, NULL},
    {7, 4, 1, rocket_f, rocket_A, rocket_B, rocket_F, NULL},
    {8, 3, 10, ss_f, ss_A, ss_B, ss_F, NULL},
    {13, 6, 1, ff_f, ff_A, ff_B, ff_F, ff_action},
};
#define N_MODELS ((int)(sizeof(MODELS) / sizeof(MODELS[0])))

/* ------------------------------------------------------------------------ */
/* Helpers restating src/utils/helper.jl                                     */
/* ------------------------------------------------------------------------ */

/* Julia LinRange(a,b,n)[j] (0-based j): lerp form used by Base (range.jl
 * `lerpi`): (1 - j/(n-1))*a + (j/(n-1))*b ; scp.jl:147, discretization.jl:197 */
static double linrange(double a, double b, int n, int j)
{
    if (n == 1) return a;
    double tt = (double)j / (double)(n - 1);
    return (1.0 - tt) * a + tt * b;
}

/* helper.jl:84-90 get_interval on a 2-point grid; helper.jl:107-118 linterp */
static void linterp2(double t, const double *f0, const double *f1, int n, double t0, double t1, double *out)
{
    double tc = fmax(t0, fmin(t1, t)); /* saturate, helper.jl:112 */
    /* k = sum(t .> grid), min 1 -> on a 2-point grid always the single interval */
    double c = (t1 - tc) / (t1 - t0);
    for (int i = 0; i < n; i++) out[i] = c * f0[i] + (1.0 - c) * f1[i];
}

/* Phi \ I : LU with partial pivoting (Julia `\` on a square dense matrix ->
 * LAPACK getrf/getrs; discretization.jl:267).  n <= ORACLE_MAX_NX. */
static int lu_inverse(const double *M, int n, double *inv)
{
    double a[ORACLE_MAX_NX * ORACLE_MAX_NX];
    int piv[ORACLE_MAX_NX];
    memcpy(a, M, (size_t)n * n * sizeof(double));
    for (int j = 0; j < n; j++) {
        int pr = j;
        double mx = fabs(a[j + n * j]);
        for (int i = j + 1; i < n; i++)
            if (fabs(a[i + n * j]) > mx) { mx = fabs(a[i + n * j]); pr = i; }
        piv[j] = pr;
        if (mx == 0.0) return 1;
        if (pr != j)
            for (int c = 0; c < n; c++) { double tmp = a[j + n * c]; a[j + n * c] = a[pr + n * c]; a[pr + n * c] = tmp; }
        for (int i = j + 1; i < n; i++) {
            a[i + n * j] /= a[j + n * j];
            double l = a[i + n * j];
            for (int c = j + 1; c < n; c++) a[i + n * c] -= l * a[j + n * c];
        }
    }
    for (int c = 0; c < n; c++) {
        double b[ORACLE_MAX_NX];
        for (int i = 0; i < n; i++) b[i] = (i == c) ? 1.0 : 0.0;
        for (int j = 0; j < n; j++) { double tmp = b[j]; b[j] = b[piv[j]]; b[piv[j]] = tmp; }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < i; j++) b[i] -= a[i + n * j] * b[j];
        for (int i = n - 1; i >= 0; i--) {
            for (int j = i + 1; j < n; j++) b[i] -= a[i + n * j] * b[j];
            b[i] /= a[i + n * i];
        }
        for (int i = 0; i < n; i++) inv[i + n * c] = b[i];
    }
    return 0;
}

/* C = A(m x k) * B(k x n), all column-major */
static void matmul(const double *A, const double *B, double *C, int m, int k, int n)
{
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++) {
            double a = 0;
            for (int l = 0; l < k; l++) a += A[i + m * l] * B[l + k * j];
            C[i + m * j] = a;
        }
}

/* ------------------------------------------------------------------------ */
/* discretize! (FOH)                                                         */
/* ------------------------------------------------------------------------ */

typedef struct {
    const oracle_model *m;
    const double *par;
    const double *p;
    const double *u0, *u1; /* ud[:,k], ud[:,k+1] */
    double t0, t1;         /* t_grid[k:k+1] */
    int k;                 /* 1-based interval */
    /* offsets into V (DiscretizationIndices, discretization.jl:113-144) */
    int ox, oA, oBm, oBp, oF, or_, oE, len;
    /* scratch */
    double *scr;
} derivs_ctx;

/* derivs_foh, discretization.jl:235-286 */
static void derivs_foh(double t, const double *V, double *dV, derivs_ctx *c)
{
    const oracle_model *m = c->m;
    int nx = m->nx, nu = m->nu, np = m->np;
    const double *x = V + c->ox;
    const double *Phi = V + c->oA;
    double *u = c->scr;             /* nu */
    double *f = u + nu;             /* nx */
    double *A = f + nx;             /* nx*nx */
    double *B = A + nx * nx;        /* nx*nu */
    double *F = B + nx * nu;        /* nx*np */
    double *Bm = F + nx * np;       /* nx*nu */
    double *Bp = Bm + nx * nu;      /* nx*nu */
    double *r = Bp + nx * nu;       /* nx */
    double *iPhi = r + nx;          /* nx*nx */
    double *E = iPhi + nx * nx;     /* nx*nx */

    linterp2(t, c->u0, c->u1, nu, c->t0, c->t1, u);     /* :249 */
    double sm = (c->t1 - t) / (c->t1 - c->t0);           /* :252 */
    double sp = (t - c->t0) / (c->t1 - c->t0);           /* :253 */
    m->f(t, c->k, x, u, c->p, c->par, f);                /* :256-259 */
    m->A(t, c->k, x, u, c->p, c->par, A);
    m->B(t, c->k, x, u, c->p, c->par, B);
    if (np > 0) m->F(t, c->k, x, u, c->p, c->par, F);
    for (int i = 0; i < nx * nu; i++) { Bm[i] = sm * B[i]; Bp[i] = sp * B[i]; } /* :260-261 */
    for (int i = 0; i < nx; i++) {                        /* r = f - A x - B u - F p  :262 */
        double a = f[i];
        for (int j = 0; j < nx; j++) a -= A[i + nx * j] * x[j];
        for (int j = 0; j < nu; j++) a -= B[i + nx * j] * u[j];
        for (int j = 0; j < np; j++) a -= F[i + nx * j] * c->p[j];
        r[i] = a;
    }
    for (int i = 0; i < nx * nx; i++) E[i] = 0.0;         /* E = I(nx), scp.jl:149 */
    for (int i = 0; i < nx; i++) E[i + nx * i] = 1.0;
    lu_inverse(Phi, nx, iPhi);                            /* :267 */
    memcpy(dV + c->ox, f, nx * sizeof(double));           /* :275-283 */
    matmul(A, Phi, dV + c->oA, nx, nx, nx);               /* dPhidt = A*Phi  :268 */
    matmul(iPhi, Bm, dV + c->oBm, nx, nx, nu);            /* :269 */
    matmul(iPhi, Bp, dV + c->oBp, nx, nx, nu);            /* :270 */
    if (np > 0) matmul(iPhi, F, dV + c->oF, nx, nx, np);  /* :271 */
    matmul(iPhi, r, dV + c->or_, nx, nx, 1);              /* :272 */
    matmul(iPhi, E, dV + c->oE, nx, nx, nx);              /* :273 */
}

/* derivs_impulse, discretization.jl:304-340: V = [x; Phi; int iPhi F; int iPhi r; int iPhi E] (no B blocks,
 * DiscretizationIndices :126-143 for IMPULSE), u0 = 0 between the nodes */
static void derivs_impulse(double t, const double *V, double *dV, derivs_ctx *c)
{
    const oracle_model *m = c->m;
    int nx = m->nx, nu = m->nu, np = m->np;
    const double *x = V + c->ox;
    const double *Phi = V + c->oA;
    double *u = c->scr;             /* nu */
    double *f = u + nu;             /* nx */
    double *A = f + nx;             /* nx*nx */
    double *F = A + nx * nx;        /* nx*np */
    double *r = F + nx * np;        /* nx */
    double *iPhi = r + nx;          /* nx*nx */
    double *E = iPhi + nx * nx;     /* nx*nx */
    for (int i = 0; i < nu; i++) u[i] = 0.0;              /* coasting, :321 */
    m->f(t, c->k, x, u, c->p, c->par, f);                 /* :326-328 */
    m->A(t, c->k, x, u, c->p, c->par, A);
    if (np > 0) m->F(t, c->k, x, u, c->p, c->par, F);
    for (int i = 0; i < nx; i++) {                        /* r = f - A x - F p  :329 */
        double a = f[i];
        for (int j = 0; j < nx; j++) a -= A[i + nx * j] * x[j];
        for (int j = 0; j < np; j++) a -= F[i + nx * j] * c->p[j];
        r[i] = a;
    }
    for (int i = 0; i < nx * nx; i++) E[i] = 0.0;
    for (int i = 0; i < nx; i++) E[i + nx * i] = 1.0;
    lu_inverse(Phi, nx, iPhi);                            /* :334 */
    memcpy(dV + c->ox, f, nx * sizeof(double));
    matmul(A, Phi, dV + c->oA, nx, nx, nx);               /* :335 */
    if (np > 0) matmul(iPhi, F, dV + c->oF, nx, nx, np);  /* :336 */
    matmul(iPhi, r, dV + c->or_, nx, nx, 1);              /* :337 */
    matmul(iPhi, E, dV + c->oE, nx, nx, nx);              /* :338 */
}

typedef void (*derivs_fn)(double, const double *, double *, derivs_ctx *);
static derivs_fn g_derivs = 0;   /* selected by oracle_discretize_method (single-threaded test infrastructure) */
#define derivs_foh_or_impulse(t, X, k, c) (g_derivs ? g_derivs : derivs_foh)((t), (X), (k), (c))

/* rk4_core_step, helper.jl:411-424 */
static void rk4_core_step(double *X, double t, double tp, derivs_ctx *c, double *w)
{
    int n = c->len;
    double h = tp - t;
    double *k1 = w, *k2 = w + n, *k3 = w + 2 * n, *k4 = w + 3 * n, *tmp = w + 4 * n;
    derivs_foh_or_impulse(t, X, k1, c);
    for (int i = 0; i < n; i++) tmp[i] = X[i] + h / 2 * k1[i];
    derivs_foh_or_impulse(t + h / 2, tmp, k2, c);
    for (int i = 0; i < n; i++) tmp[i] = X[i] + h / 2 * k2[i];
    derivs_foh_or_impulse(t + h / 2, tmp, k3, c);
    for (int i = 0; i < n; i++) tmp[i] = X[i] + h * k3[i];
    derivs_foh_or_impulse(t + h, tmp, k4, c);
    for (int i = 0; i < n; i++) X[i] = X[i] + h / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}

/*
 * oracle_discretize: discretize!, discretization.jl:160-217 (FOH branch) +
 * set_update_matrices :354-406.  One problem.  t_grid = LinRange(0,1,N)
 * (scp.jl:147).  Outputs column-major: A[nx,nx,N-1], Bm/Bp[nx,nu,N-1],
 * F[nx,np,N-1], r[nx,N-1], E[nx,nx,N-1], defect[nx,N-1]; *feas (0/1).
 * Returns 0 ok, 1 bad model id.
 */
int oracle_discretize(int model_id, const double *par, int N, int Nsub,
                      const double *xd, const double *ud, const double *p,
                      const double *iSx_diag, double feas_tol,
                      double *A, double *Bm, double *Bp, double *F, double *r, double *E,
                      double *defect, int *feas)
{
    if (model_id < 0 || model_id >= N_MODELS) return 1;
    const oracle_model *m = &MODELS[model_id];
    int nx = m->nx, nu = m->nu, np = m->np;
    derivs_ctx c;
    c.m = m; c.par = par; c.p = p;
    c.ox = 0; c.oA = nx; c.oBm = c.oA + nx * nx; c.oBp = c.oBm + nx * nu;
    c.oF = c.oBp + nx * nu; c.or_ = c.oF + nx * np; c.oE = c.or_ + nx; c.len = c.oE + nx * nx;
    int len = c.len;
    double *V = (double *)calloc((size_t)len * 6 + 8 * ORACLE_MAX_NX * ORACLE_MAX_NX + nx * np + 64, sizeof(double));
    double *w = V + len;
    c.scr = w + 5 * len;
    *feas = 1;                                                  /* :179 */
    for (int k = 1; k <= N - 1; k++) {                          /* :182 */
        memset(V, 0, len * sizeof(double));                     /* V0 :177 */
        for (int i = 0; i < nx; i++) V[c.oA + i + nx * i] = 1.0; /* :178 */
        memcpy(V + c.ox, xd + (size_t)nx * (k - 1), nx * sizeof(double)); /* :185 */
        c.k = k;
        c.t0 = linrange(0.0, 1.0, N, k - 1);
        c.t1 = linrange(0.0, 1.0, N, k);
        c.u0 = ud + (size_t)nu * (k - 1);
        c.u1 = ud + (size_t)nu * k;
        for (int j = 1; j < Nsub; j++) {                        /* rk4_generic, helper.jl:483-498 */
            double ta = linrange(c.t0, c.t1, Nsub, j - 1);      /* :197 */
            double tb = linrange(c.t0, c.t1, Nsub, j);
            rk4_core_step(V, ta, tb, &c, w);
            if (m->action) m->action(V + c.ox);                 /* helper.jl:494-496 */
        }
        /* set_update_matrices :381-403 */
        const double *Ak = V + c.oA;
        memcpy(A + (size_t)nx * nx * (k - 1), Ak, nx * nx * sizeof(double));
        matmul(Ak, V + c.oBm, Bm + (size_t)nx * nu * (k - 1), nx, nx, nu);
        matmul(Ak, V + c.oBp, Bp + (size_t)nx * nu * (k - 1), nx, nx, nu);
        if (np > 0) matmul(Ak, V + c.oF, F + (size_t)nx * np * (k - 1), nx, nx, np);
        matmul(Ak, V + c.or_, r + (size_t)nx * (k - 1), nx, nx, 1);
        matmul(Ak, V + c.oE, E + (size_t)nx * nx * (k - 1), nx, nx, nx);
        /* defect :205-210 */
        double nrm = 0;
        for (int i = 0; i < nx; i++) {
            double d = xd[(size_t)nx * k + i] - V[c.ox + i];
            defect[(size_t)nx * (k - 1) + i] = d;
            nrm = fmax(nrm, fabs(iSx_diag[i] * d));
        }
        if (nrm > feas_tol) *feas = 0;
    }
    free(V);
    return 0;
}

/*
 * IMPULSE branch of discretize! (discretization.jl:186-193) + set_update_matrices (:384-390): the state is
 * impulse-updated at the node, x_k+ = x_k + f(t_k, -k, x_k, u_k, p), the system coasts to the next node, and
 * B_k = A_k * B(t_k, -k, x_k, u_k, p).  Output Bm = B_k (dyn.B[1]); Bp is zeroed (the reference's DLTV has one B).
 */
int oracle_discretize_impulse(int model_id, const double *par, int N, int Nsub,
                              const double *xd, const double *ud, const double *p,
                              const double *iSx_diag, double feas_tol,
                              double *A, double *Bm, double *Bp, double *F, double *r, double *E,
                              double *defect, int *feas)
{
    if (model_id < 0 || model_id >= N_MODELS) return 1;
    const oracle_model *m = &MODELS[model_id];
    int nx = m->nx, nu = m->nu, np = m->np;
    derivs_ctx c;
    c.m = m; c.par = par; c.p = p;
    c.ox = 0; c.oA = nx; c.oBm = c.oBp = -1;
    c.oF = c.oA + nx * nx; c.or_ = c.oF + nx * np; c.oE = c.or_ + nx; c.len = c.oE + nx * nx;
    int len = c.len;
    double *V = (double *)calloc((size_t)len * 6 + 8 * ORACLE_MAX_NX * ORACLE_MAX_NX + nx * np + 64, sizeof(double));
    double *w = V + len;
    double fimp[ORACLE_MAX_NX], Btk[ORACLE_MAX_NX * ORACLE_MAX_NX];
    c.scr = w + 5 * len;
    *feas = 1;
    g_derivs = derivs_impulse;
    for (int k = 1; k <= N - 1; k++) {
        memset(V, 0, len * sizeof(double));
        for (int i = 0; i < nx; i++) V[c.oA + i + nx * i] = 1.0;
        c.k = k;
        c.t0 = linrange(0.0, 1.0, N, k - 1);
        c.t1 = linrange(0.0, 1.0, N, k);
        const double *xk = xd + (size_t)nx * (k - 1), *uk = ud + (size_t)nu * (k - 1);
        m->f(c.t0, -k, xk, uk, p, par, fimp);                                  /* :191 */
        for (int i = 0; i < nx; i++) V[c.ox + i] = xk[i] + fimp[i];            /* :192 */
        for (int j = 1; j < Nsub; j++) {
            double ta = linrange(c.t0, c.t1, Nsub, j - 1);
            double tb = linrange(c.t0, c.t1, Nsub, j);
            rk4_core_step(V, ta, tb, &c, w);
            if (m->action) m->action(V + c.ox);
        }
        const double *Ak = V + c.oA;
        memcpy(A + (size_t)nx * nx * (k - 1), Ak, nx * nx * sizeof(double));
        m->B(c.t0, -k, xk, uk, p, par, Btk);                                   /* :388 */
        matmul(Ak, Btk, Bm + (size_t)nx * nu * (k - 1), nx, nx, nu);           /* B_k = A_k * Btk  :389 */
        memset(Bp + (size_t)nx * nu * (k - 1), 0, (size_t)nx * nu * sizeof(double));
        if (np > 0) matmul(Ak, V + c.oF, F + (size_t)nx * np * (k - 1), nx, nx, np);
        matmul(Ak, V + c.or_, r + (size_t)nx * (k - 1), nx, nx, 1);
        matmul(Ak, V + c.oE, E + (size_t)nx * nx * (k - 1), nx, nx, nx);
        double nrm = 0;
        for (int i = 0; i < nx; i++) {
            double d = xd[(size_t)nx * k + i] - V[c.ox + i];
            defect[(size_t)nx * (k - 1) + i] = d;
            nrm = fmax(nrm, fabs(iSx_diag[i] * d));
        }
        if (nrm > feas_tol) *feas = 0;
    }
    g_derivs = 0;
    free(V);
    return 0;
}

/* Batched convenience wrapper (sequential loop, like the reference's
 * `for trial` loop): arrays carry a trailing batch dimension. */
int oracle_discretize_batch(int model_id, const double *par, int N, int Nsub, int batch,
                            const double *xd, const double *ud, const double *p,
                            const double *iSx_diag, double feas_tol,
                            double *A, double *Bm, double *Bp, double *F, double *r, double *E,
                            double *defect, int *feas)
{
    if (model_id < 0 || model_id >= N_MODELS) return 1;
    const oracle_model *m = &MODELS[model_id];
    size_t nx = m->nx, nu = m->nu, np = m->np, M = N - 1;
    for (int b = 0; b < batch; b++) {
        int rc = oracle_discretize(model_id, par, N, Nsub, xd + nx * N * b, ud + nu * N * b, p + np * b,
                                   iSx_diag, feas_tol, A + nx * nx * M * b, Bm + nx * nu * M * b,
                                   Bp + nx * nu * M * b, F + nx * np * M * b, r + nx * M * b,
                                   E + nx * nx * M * b, defect + nx * M * b, feas + b);
        if (rc) return rc;
    }
    return 0;
}

/* Direct model access for Jacobian finite-difference checks in the tests. */
/* ------------------------------------------------------------------------ */
/* propagate (FOH), discretization.jl:515-541                                 */
/* ------------------------------------------------------------------------ */

/* get_interval, helper.jl:84-90 (1-based bin: number of grid points strictly below x, at least 1) */
static int get_interval(double x, const double *grid, int n)
{
    int k = 0;
    for (int i = 0; i < n; i++) if (x > grid[i]) k++;
    if (k == 0) k = 1;
    return k;
}

/* linterp on the N-point grid, helper.jl:107-118 (f_cps column-major [n, N]) */
static void linterp_grid(double t, const double *f_cps, const double *grid, int n, int N, double *out)
{
    if (t < grid[0]) t = grid[0];
    if (t > grid[N - 1]) t = grid[N - 1];
    int k = get_interval(t, grid, N); /* 1-based: uses columns k and k+1 */
    double c = (grid[k] - t) / (grid[k] - grid[k - 1]);
    for (int i = 0; i < n; i++) out[i] = c * f_cps[i + n * (k - 1)] + (1 - c) * f_cps[i + n * k];
}

/*
 * oracle_propagate: propagate(sol, pbm; res) for the FOH method.  Integrates the nonlinear dynamics from
 * xd[:,1] over tc = LinRange(0,1,res) with u(t) = linterp of ud on t_grid (Trajectory(td, ud, :linear), :531)
 * and classic RK4 steps between consecutive tc (rk4(...; full=true), helper.jl:483-498).  The node index the
 * reference passes to f is k(t) = max(floor(t/(N-1))+1, N) = N for every t (:529).
 * xc is [nx, res] column-major.  Returns 0 ok, 1 bad model id.
 */
int oracle_propagate(int model_id, const double *par, int N, const double *xd, const double *ud, const double *p,
                     int res, double *xc)
{
    if (model_id < 0 || model_id >= N_MODELS) return 1;
    const oracle_model *m = &MODELS[model_id];
    int nx = m->nx, nu = m->nu;
    double *grid = (double *)malloc(sizeof(double) * (size_t)N);
    for (int j = 0; j < N; j++) grid[j] = linrange(0.0, 1.0, N, j);
    double x[ORACLE_MAX_NX], k1[ORACLE_MAX_NX], k2[ORACLE_MAX_NX], k3[ORACLE_MAX_NX], k4[ORACLE_MAX_NX], tmp[ORACLE_MAX_NX], u[ORACLE_MAX_NX];
    for (int i = 0; i < nx; i++) { x[i] = xd[i]; xc[i] = x[i]; }
    for (int j = 1; j < res; j++) {
        double t = linrange(0.0, 1.0, res, j - 1), tp = linrange(0.0, 1.0, res, j), h = tp - t;
        linterp_grid(t, ud, grid, nu, N, u); m->f(t, N, x, u, p, par, k1);
        for (int i = 0; i < nx; i++) tmp[i] = x[i] + h / 2 * k1[i];
        linterp_grid(t + h / 2, ud, grid, nu, N, u); m->f(t + h / 2, N, tmp, u, p, par, k2);
        for (int i = 0; i < nx; i++) tmp[i] = x[i] + h / 2 * k2[i];
        m->f(t + h / 2, N, tmp, u, p, par, k3);
        for (int i = 0; i < nx; i++) tmp[i] = x[i] + h * k3[i];
        linterp_grid(t + h, ud, grid, nu, N, u); m->f(t + h, N, tmp, u, p, par, k4);
        for (int i = 0; i < nx; i++) x[i] = x[i] + h / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        if (m->action) m->action(x);                          /* actions = pbm.traj.integ_actions, :537 */
        for (int i = 0; i < nx; i++) xc[i + nx * j] = x[i];
    }
    free(grid);
    return 0;
}

int oracle_model_dims(int model_id, int *nx, int *nu, int *np)
{
    if (model_id < 0 || model_id >= N_MODELS) return 1;
    *nx = MODELS[model_id].nx; *nu = MODELS[model_id].nu; *np = MODELS[model_id].np;
    return 0;
}
int oracle_model_eval(int model_id, const double *par, double t, int k, const double *x, const double *u,
                      const double *p, double *f, double *A, double *B, double *F)
{
    if (model_id < 0 || model_id >= N_MODELS) return 1;
    const oracle_model *m = &MODELS[model_id];
    m->f(t, k, x, u, p, par, f);
    m->A(t, k, x, u, p, par, A);
    m->B(t, k, x, u, p, par, B);
    if (m->np > 0) m->F(t, k, x, u, p, par, F);
    return 0;
}
