/*
 * scp_conic.h -- C ABI of the generic batched conic solver (MI355X).
 *
 * Drop-in for the reference's one true plugin seam, the convex solver behind `ConicProgram`:
 *
 *   set_optimizer(mdl, solver.Optimizer)      src/parser/program.jl:63-76   (pars.solver, src/solvers/ptr.jl:69)
 *   solve!(prg) = JuMP.optimize!(mdl)         src/parser/program.jl:419-424
 *   termination_status / value / objective    src/parser/program.jl:427-431, src/parser/block.jl:368-394
 *
 * which hands ECOS (libecos, `ECOS_setup` / `ECOS_solve`) the standard form
 *
 *     min 1/2 x'Px + c'x   s.t.   A x = b,   G x + s = h,   s in K = R+^l x K_0 x ... x K_{ncones-1}
 *     K_c = Q^{q[c]} (second-order cone, q[c] >= 1)  or, for q[c] = -3, the EXPONENTIAL cone
 *           {(x, y, w): y exp(x / y) <= w, y > 0}  (3 rows in that order; src/parser/cone.jl:45 EXP = MOI.ExponentialCone;
 *           GuSTO's softplus penalty, src/solvers/gusto.jl:996-1031).  m = l + sum |q[c]|.
 *           Limit: the method is an infeasible-start one without ECOS's self-dual embedding; a program whose exponential cone can
 *           only be entered against its curvature (a CONSTANT x row >= ~1.5 y) ends ITERATION_LIMIT instead of OPTIMAL
 *           (tests/test_oracle_exp_cone.py pins this).  The softplus cones have the penalty variable in the x row and are not of that kind.
 *
 * (P = 0 in ECOS; the quadratic term is accepted natively here instead of going through MOI's quadratic->SOC
 * bridge).  The entry points mirror the shape of ECOS's C API -- sparse matrices in compressed-column form, cone
 * dimensions l and q[] -- for a BATCH of programs that share one sparsity pattern (every subproblem of a Monte-Carlo
 * SCP batch, every SCP iteration): the pattern is analysed once (`scp_conic_create` ~ ECOS_setup), values are passed
 * per problem with a trailing batch dimension (`scp_conic_solve_batch_host` ~ ECOS_solve), column-major like Julia.
 * Callers on the reference side: `solve_subproblem!` (src/solvers/scp.jl:942-950), `correct_convex!` (:275-361),
 * `compute_scaling` (:376-517).
 *
 * Conventions: plain C, fp64, 0-based indices, integer return codes of scp_mi355x.h (0 = ok), nothing throws.
 */
#ifndef SCP_CONIC_H
#define SCP_CONIC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct scp_conic *scp_conic_handle;

/* per-problem exit status (MOI.TerminationStatusCode subset the reference inspects, src/solvers/scp.jl:965-980,467-473) */
typedef enum {
    SCP_CONIC_OPTIMAL = 0,
    SCP_CONIC_ALMOST_OPTIMAL = 1,
    SCP_CONIC_ITERATION_LIMIT = 2,
    SCP_CONIC_NUMERICAL_ERROR = 3,
    SCP_CONIC_INFEASIBLE = 4,        /* primal infeasible (certificate found) */
    SCP_CONIC_DUAL_INFEASIBLE = 5    /* unbounded (certificate found)         */
} scp_conic_status;

/* solver options; defaults = ECOS's (feastol = abstol = reltol = 1e-8, maxit = 100) */
typedef struct {
    int max_iter;
    double feastol, abstol, reltol;
    double reg;        /* static regularisation of the KKT matrix (ECOS: delta); < 0 (default): chosen from the   */
                       /* pattern -- 1e-10 (pure LPs and programs with fewer cone rows than variables: 1e-8); 1e-6 when variables appear in no cone row and have no quadratic cost */
    double dyn_eps;    /* a pivot with sign * D <= dyn_eps is replaced by sign * dyn_delta (ECOS's rule    */
    double dyn_delta;  /* and constants: 1e-13, 2e-7)                                                       */
    int nref;          /* max iterative-refinement steps per Newton solve                          */
    double ref_tol;    /* refinement stops at |res|_2 <= ref_tol (1 + |rhs|_2)                     */
    double step;       /* fraction of the step to the cone boundary                                */
} scp_conic_opts;

void scp_conic_default_opts(scp_conic_opts *o);

/* bits of `shared_mask`: the array is ONE copy used by every problem of the batch (no batch dimension) */
#define SCP_CONIC_SHARED_C 1u
#define SCP_CONIC_SHARED_B 2u
#define SCP_CONIC_SHARED_H 4u
#define SCP_CONIC_SHARED_G 8u
#define SCP_CONIC_SHARED_A 16u
#define SCP_CONIC_SHARED_P 32u

/*
 * Analyse a sparsity pattern (~ ECOS_setup): n variables, p equality rows, m cone rows = l + sum(q).
 * P: upper triangle of the n x n cost matrix (Pp may describe an empty matrix), A: p x n, G: m x n; CSC with sorted
 * row indices.  perm: optional fill-reducing ordering of the (n+p+m) KKT unknowns [x; y; z] (NULL: minimum degree).
 * batch_capacity: most problems per solve call.
 */
int scp_conic_create(int n, int p, int m, int l, int ncones, const int *q, const int *Pp, const int *Pi,
                     const int *Ap, const int *Ai, const int *Gp, const int *Gi, const int *perm,
                     int batch_capacity, int device, scp_conic_handle *out);
int scp_conic_destroy(scp_conic_handle h);
const char *scp_conic_last_error(scp_conic_handle h);

/* symbolic statistics: stats[0] = nnz(L), [1] = multiply-adds per numeric factorisation, [2] = KKT dimension,
 * [3] = nnz(Gt) (cone rows unioned per column), [4] = device bytes per problem, [5] / [6] = elimination levels of the
 * factorisation / of the backward substitution (barriers per sweep), [7] = worker waves per group of 64 problems,
 * [8] = nested-dissection depth of the ordering in use (0: sequential minimum degree), [9] = problems re-solved so far by
 * the sequential fallback schedule, [10] = problems solved so far, [11] = elimination levels of the fallback schedule
 * (0: none kept or switched off), [12] = problems the fallback pass left with a usable solution or a certificate,
 * [13..15] = 0 (reserved).  Ordering: SCP_CONIC_ORDER = auto (default: the cheaper of the sequential minimum-degree order and
 * the nested dissection of the chain of node blocks, priced for the batch capacity's launch geometry; a dissection keeps
 * the sequential schedule as a per-problem fallback for ITERLIM / NUMERR exits, dropped once it stops rescuing them, made
 * the primary schedule when it rescues most of a launch) | nd | seq. */
int scp_conic_stats(scp_conic_handle h, long long stats[16]);

/*
 * Solve B programs (~ ECOS_solve).  Values: c[n,B], b[p,B], h[m,B], Gx[nnz(G),B], Ax[nnz(A),B], Px[nnz(P),B]
 * (host pointers, column-major, batch index last; arrays flagged in shared_mask have no batch dimension).
 * Results: x[n,B], y[p,B], z[m,B], s[m,B], status[B] (scp_conic_status), iters[B],
 * info[8,B] = (pcost, dcost, gap, pres, dres, relgap, dynamic regularisations, refinement steps),
 * *seconds = device time of the solve kernel.  Any output pointer may be NULL.
 */
int scp_conic_solve_batch_host(scp_conic_handle h, int B, const double *c, const double *b, const double *hvec,
                               const double *Gx, const double *Ax, const double *Px, unsigned shared_mask,
                               const scp_conic_opts *opts, double *x, double *y, double *z, double *s,
                               int32_t *status, int32_t *iters, double *info, double *seconds);

/*
 * One-shot convenience with the signature sketched in SURVEY.md section 8(b): analyse + solve + destroy.
 * No quadratic term; every value array carries the batch dimension.
 */
int socp_solve_batch(int n, int m, int p, int l, int ncones, const int *q, const int *Gp, const int *Gi,
                     const double *Gx, const int *Ap, const int *Ai, const double *Ax, const double *c,
                     const double *hvec, const double *b, int B, double *x, double *y, double *s, double *z,
                     int32_t *status);

#ifdef __cplusplus
}
#endif
#endif /* SCP_CONIC_H */
