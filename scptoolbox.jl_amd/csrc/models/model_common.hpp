// Shared helpers for the compiled device models that replace the reference's
// user closures traj.f/A/B/F (src/parser/problem.jl:432-450).
#pragma once
#include <hip/hip_runtime.h>

#define SCP_DEV __host__ __device__ __forceinline__

namespace scp {

template <int N>
SCP_DEV void zero(double (&a)[N])
{
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = 0.0;
}

// Traits every model inherits; a model overrides what differs.
//
// PARAMETER VECTOR.  p = [global (M::np); node parameters (M::np_node, N) column-major], length np + np_node N -- the
// free-flyer's p = [t_f; delta(6, N)] (test/examples/freeflyer/parameters.jl:121-128).  The dynamics, the
// boundary conditions and the parameter-only rows see the GLOBAL parameters only; the constraints of node k (X rows,
// s) additionally see that node's own np_node parameters.  Their parameter Jacobians are COMPACT: npc = np + np_node
// columns, column j < np = global parameter j, column np + i = node parameter i of node k, i.e. entry
// np + np_node (k - 1) + i of p.
struct ModelDefaults {
    static constexpr int np_node = 0;
    // the parameter-only rows (glin_rows) are members of the convex STATE set X (soft under GuSTO, repeated at every
    // node: freeflyer/definition.jl:318-331) instead of the input set U (hard, kept once: quadrotor/definition.jl:223-250)
    static constexpr bool global_rows_in_X = false;
    // the first linf_groups * linf_rows linear rows are LINF cones lowered to rows (MOI's NormInfinity bridge); the rows
    // of one cone share one cone indicator under GuSTO (src/parser/problem.jl:744-763)
    static constexpr int linf_groups = 0, linf_rows = 0;
    // s(t, k, x, p) does not depend on the input: the model is admissible for GuSTO (gusto.jl:757-792)
    static constexpr bool s_input_free = false;
    // models with state-dependent Jacobians: largest PHYSICAL RK4 step time_dilation(p) * h [s] up to which the variational
    // form of discretize! (K1x, discretize_kernel.hpp) agrees with the reference formulation to 1e-10; 0 = never
    static constexpr double var_form_max_phys_step = 0.0;
    // a model may provide out = A(x, p) v without forming A (Amulx): used by K1x instead of the dense product
    static constexpr bool has_amulx = false;
    // structure of the state-transition matrix Phi the reference-form kernel K1 may rely on (discretize_kernel.hpp, the cooperative LU):
    // the first lu_lead columns of A(t, x, u, p) have NO entry on or below the diagonal (states nothing in rows >= their own index
    // depends on), so Phi[:, s] keeps a unit diagonal and exact zeros below it for s < lu_lead; lu_decoupled: the rows < lu_lead of A
    // are zero in the columns >= lu_lead too (block-diagonal A and Phi).  0 / false: no assumption.
    static constexpr int lu_lead = 0;
    static constexpr bool lu_decoupled = false;
    template <class PP>
    SCP_DEV static double time_dilation(const PP&, const double*) { return 0.0; }
};

template <class M>
SCP_DEV constexpr int np_total(int N)
{
    return M::np + M::np_node * N;
}
template <class M>
SCP_DEV constexpr int np_compact()
{
    return M::np + M::np_node;
}

// Cone indicators q of the convex state set X at node k (1-based) -- the numerical mode of define_conic_constraint!
// (src/parser/problem.jl:783-803): q <= 0 iff the point is in the cone.  Derived from the model's own rows with the
// classification the host formulation uses (subproblem.py::split_state_rows / state_cones): second-order cones and
// linear rows WITHOUT an input column belong to X; the lowered rows of one LINF cone share one indicator (their
// maximum = |z_1..|_inf - z_0); parameter-only rows count at every node when M::global_rows_in_X.
//   count: nq = (#X cones) + (global_rows_in_X ? ng : 0) + linf_groups + (#single X rows)
// f(q) is called once per indicator (order irrelevant to the callers: sums and maxima).
template <class M, class F>
SCP_DEV void for_each_x_indicator(const typename M::Params& P, double t, int k, const double* x, const double* p, int N, F&& f)
{
    constexpr int nx = M::nx, nu = M::nu, nz = nx + nu, np = M::np, npc = np_compact<M>(), npca = npc > 0 ? npc : 1;
    const double* pk = p + np + (long)M::np_node * (k - 1);
    (void)N;
    if constexpr (M::nsoc > 0) {
        double Mm[M::nsoc * 4 * nz], m[M::nsoc * 4];
        M::soc_rows(P, t, k, Mm, m);
        for (int c = 0; c < M::nsoc; c++) {
            bool has_u = false;
            for (int r = 0; r < 4; r++) for (int j = nx; j < nz; j++) has_u = has_u || Mm[(4 * c + r) * nz + j] != 0.0;
            if (has_u) continue;
            double z[4];
            for (int r = 0; r < 4; r++) {
                double a = m[4 * c + r];
                for (int j = 0; j < nx; j++) a += Mm[(4 * c + r) * nz + j] * x[j];
                z[r] = a;
            }
            f(sqrt(z[1] * z[1] + z[2] * z[2] + z[3] * z[3]) - z[0]);
        }
    }
    if constexpr (M::ng > 0 && M::global_rows_in_X) {
        double Lg[M::ng * (np > 0 ? np : 1)], lg[M::ng];
        M::glin_rows(P, Lg, lg);
        for (int i = 0; i < M::ng; i++) {
            double a = lg[i];
            for (int j = 0; j < np; j++) a += Lg[i * np + j] * p[j];
            f(a);
        }
    }
    if constexpr (M::nl > 0) {
        double L[M::nl * nz], Lp[M::nl * npca], l[M::nl];
        for (int i = 0; i < M::nl * npca; i++) Lp[i] = 0.0;
        M::lin_rows(P, t, k, L, Lp, l);
        auto value = [&](int i, bool& is_x) {
            bool has_u = false, any = false;
            for (int j = nx; j < nz; j++) has_u = has_u || L[i * nz + j] != 0.0;
            for (int j = 0; j < nx; j++) any = any || L[i * nz + j] != 0.0;
            for (int j = 0; j < npc; j++) any = any || Lp[i * npca + j] != 0.0;
            is_x = !has_u && any;
            double a = l[i];
            for (int j = 0; j < nx; j++) a += L[i * nz + j] * x[j];
            for (int j = 0; j < np; j++) a += Lp[i * npca + j] * p[j];
            for (int j = 0; j < M::np_node; j++) a += Lp[i * npca + np + j] * pk[j];
            return a;
        };
        for (int g = 0; g < M::linf_groups; g++) {
            double q = -1e300;
            bool is_x = true;
            for (int r = 0; r < M::linf_rows; r++) { bool ix; const double v = value(g * M::linf_rows + r, ix); q = fmax(q, v); is_x = is_x && ix; }
            if (is_x) f(q);
        }
        for (int i = M::linf_groups * M::linf_rows; i < M::nl; i++) {
            bool is_x;
            const double v = value(i, is_x);
            if (is_x) f(v);
        }
    }
}

// number of cone indicators for_each_x_indicator visits (host side, scp_model_query)
template <class M>
static int count_x_indicators(const typename M::Params& P, int N)
{
    double x[M::nx];
    for (int i = 0; i < M::nx; i++) x[i] = 0.0;
    const int npt = np_total<M>(N);
    double* p = new double[npt > 0 ? npt : 1];
    for (int i = 0; i < npt; i++) p[i] = 0.0;
    int n = 0;
    for_each_x_indicator<M>(P, 0.0, 1, x, p, N, [&](double) { n++; });
    delete[] p;
    return n;
}

}  // namespace scp
