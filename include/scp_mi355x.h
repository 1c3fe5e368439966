/*
 * scp_mi355x.h -- C ABI of the MI355X-native SCP inner loop.
 *
 * Drop-in boundary for the two functions on the hot path of UW-ACL/SCPToolbox.jl
 * (paths relative to the reference root):
 *
 *   discretize!(ref, pbm)                 src/solvers/discretization.jl:160-217
 *   solve_subproblem!(spbm, constructor)  src/solvers/scp.jl:942-950
 *
 * plus the PTR iteration that strings them together (src/solvers/ptr.jl:448-532)
 * so that a Monte-Carlo batch can stay resident in HBM between iterations.
 *
 * Conventions
 *  - plain C, no torch types; all arrays are IEEE fp64, COLUMN-MAJOR exactly as
 *    Julia lays out Array{Float64} (a Julia array can be passed with `pointer`).
 *    A trailing batch dimension B is appended to every per-problem array.
 *  - every entry point returns an scp_status (0 = ok); nothing throws or aborts
 *    across this boundary (mirrors SCPStatus, src/utils/globals.jl:34-42).
 *  - `scp_handle` owns device scratch and one HIP stream; it is re-entrant per
 *    handle, there is no global state.  Caller owns every buffer it passes.
 *  - *_host entry points take host pointers (copies in/out, PCIe inclusive);
 *    *_dev entry points take device pointers and are asynchronous on the
 *    handle's stream (call scp_sync before reading results).
 *
 * The user closures f/A/B/F/s/C/D/G/gic/... of `TrajectoryProblem`
 * (src/parser/problem.jl:64-121) cannot cross an FFI; they are replaced by a
 * registry of compiled device models selected by `model_id` plus a POD
 * parameter blob.
 */
#ifndef SCP_MI355X_H
#define SCP_MI355X_H

#include <stddef.h>
#include <stdint.h>

#include "scp_conic.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct scp_problem *scp_handle;

typedef enum {
    SCP_OK = 0,
    SCP_ERR_BAD_ARGUMENT = 1,
    SCP_ERR_UNKNOWN_MODEL = 2,
    SCP_ERR_NO_DEVICE = 3,
    SCP_ERR_HIP = 4,
    SCP_ERR_ALLOC = 5,
    SCP_ERR_BATCH_TOO_LARGE = 6,
    SCP_ERR_UNSUPPORTED = 7,
    SCP_ERR_PEER = 8            /* scp_ptr_run_sharded: another rank failed inside a window (it says so through the window's all-reduce;
                                   every rank leaves the loop, no collective is left unmatched) */
} scp_status;

/* model registry (replaces traj.f/A/B/F..., src/parser/problem.jl:432-450) */
typedef enum {
    SCP_MODEL_DOUBLE_INTEGRATOR = 0, /* builder-defined, see DESIGN.md       */
    SCP_MODEL_QUADROTOR = 1,         /* test/examples/quadrotor              */
    SCP_MODEL_ROCKET_LANDING = 2,    /* builder-defined over rocket_landing  */
    SCP_MODEL_STARSHIP = 3,          /* test/examples/starship_flip          */
    SCP_MODEL_FREEFLYER = 4          /* test/examples/freeflyer: p = [t_f; delta(6, N)] (np = 1 global + np_node = 6) */
} scp_model_id;

/* DiscretizationType, src/parser/problem.jl:52 */
typedef enum { SCP_FOH = 0, SCP_IMPULSE = 1 } scp_disc_method;

/* Static description of a model (dimensions the caller needs to size buffers).
 *
 * PARAMETER VECTOR.  p = [global parameters (np); node parameters (np_node, N) column-major]: its length is
 * np + np_node * N.  Only the free-flyer has node parameters (p = [t_f; delta(6, N)], one room-SDF slack per room and node,
 * test/examples/freeflyer/parameters.jl:121-128).  Dynamics, boundary conditions and parameter-only rows see the
 * global parameters; the constraints of node k (X rows, s) additionally see that node's own np_node parameters, and their
 * parameter Jacobians are COMPACT: np + np_node columns, column j < np = global parameter j, column np + i = entry
 * np + np_node (k - 1) + i of p. */
typedef struct {
    int nx, nu, np;   /* state / input / GLOBAL parameter dims (problem_set_dims!)   */
    int npF;          /* number of structurally non-zero columns of F (F8)      */
    int Fcols[8];     /* their 0-based column indices into p                    */
    int ns;           /* rows of the non-convex path constraint s               */
    int nic, ntc;     /* rows of the initial / terminal boundary conditions     */
    int npar;         /* doubles in the shared model parameter blob             */
    int npp;          /* doubles of per-problem data (Monte-Carlo ICs)          */
    int nl, nsoc, ng; /* convex-set rows: linear, second-order cones (dim 4), p-only */
    int structured;   /* 1: the stage-structured PTR fast path (scp_ptr_*) exists for this model; 0: subproblems run  */
                      /* through the generic conic path only (scp_sub_*, scp_scvx_*, scp_gusto_*)                  */
    int has_subproblem; /* 0: only discretize! / propagate / the initial guess exist; the subproblem entry points return
                           SCP_ERR_UNSUPPORTED (no such model at present) */
    int np_node;      /* parameters per node (see above); 0 for every model but the free-flyer                      */
    int global_rows_in_X; /* 1: the parameter-only rows are members of the convex STATE set X (soft under GuSTO and repeated
                           at every node, freeflyer/definition.jl:318-331), 0: of the input set U (hard, kept once)   */
    int linf_groups, linf_rows; /* the first linf_groups * linf_rows linear rows are LINF cones lowered to rows (MOI's
                           NormInfinity bridge); the rows of one cone share one cone indicator under GuSTO             */
    int s_input_free; /* 1: s(t, k, x, p) does not depend on the input -- admissible for GuSTO (gusto.jl:757-792)      */
} scp_model_info;

/* SCPScaling, src/solvers/scp.jl:39-49 (diagonals only; the reference's
 * Sx/Su/Sp are diagonal matrices, scp.jl:489-511). */
typedef struct {
    const double *Sx, *cx; /* [nx] */
    const double *Su, *cu; /* [nu] */
    const double *Sp, *cp; /* [np + np_node N] */
} scp_scaling;

typedef struct {
    int model_id;            /* scp_model_id                                    */
    const double *model_par; /* [npar] what the reference keeps in traj.mdl: every vehicle / environment constant of the
                                model, shared by the batch (layouts: the model headers in csrc/models) */
    int N;                   /* temporal grid nodes (pars.N)                     */
    int Nsub;                /* sub-interval integration nodes (pars.Nsub)       */
    int disc_method;         /* scp_disc_method (pars.disc_method)               */
    double feas_tol;         /* dynamic feasibility tolerance (pars.feas_tol)    */
    scp_scaling scale;       /* variable scaling (pbm.common.scale)              */
    int batch_capacity;      /* max problems per call                            */
    int device;              /* HIP device ordinal                               */
} scp_problem_desc;

int scp_model_query(int model_id, scp_model_info *info);

/*
 * Host-side evaluation of a compiled model's convex path constraints and cost at node k (1-based, t_k =
 * LinRange(0,1,N)[k]) -- what the reference obtains by calling the closures traj.X / traj.U (problem_set_X!/U!,
 * src/parser/problem.jl:500-542) and the cost (problem_set_terminal_cost!/running_cost!, :553-600) inside its
 * formulation code (src/solvers/scp.jl:685-734, 552-601).  With z = [x; u], npc = np + np_node (compact parameter
 * columns: the global parameters, then node k's own):
 *   L[nl,nz], Lp[nl,npc], l[nl]  (row-major)   L z + Lp [p_glob; p_node_k] + l <= 0
 *   Mm[4 nsoc,nz], m[4 nsoc]                    Mm z + m in Q^4 per cone (first row >= norm of the other three)
 *   Lg[ng,np], lg[ng]                           Lg p_glob + lg <= 0 (parameter-only rows)
 *   cost = [Qu[nu], lu[nu], lx[nx], tx[nx], tp[npc], Qp[npc]]:  Gamma = sum Qu_i u_i^2 + lu'u + lx'x,
 *          phi = tx'x_N + sum_j tp_j p_j + Qp_j p_j^2 over the global parameters + the same with the node entries of
 *          tp / Qp over the node parameters of EVERY node (free-flyer: -eps_sdf sum(delta))
 * Any pointer may be NULL.  Pure host code (no device needed): used by the host-side subproblem formulation.
 */
int scp_model_rows(int model_id, const double *model_par, int N, int k, double *L, double *Lp, double *l,
                   double *Mm, double *m, double *Lg, double *lg, double *cost);

/*
 * Number of cone indicators of the convex state set X per node -- define_conic_constraint! in its GuSTO mode
 * (src/parser/problem.jl:686-807): one per second-order cone and per linear row without an input column, one per LINF
 * group, and the parameter-only rows when global_rows_in_X.  A GuSTO template penalises nst = *nq + ns quantities per node.
 */
int scp_model_state_indicators(int model_id, const double *model_par, int N, int *nq);

/*
 * Host-side evaluation of a compiled model's closures at node k (1-based) and the point (x[nx], u[nu], p[np + np_node N]) --
 * what Julia obtains by calling traj.f / A / B / F (src/parser/problem.jl:432-450), traj.s / C / D / G (:560-600) and the
 * numerical mode of define_conic_constraint! (:783-803): f[nx], A[nx,nx], B[nx,nu] (column-major), F[nx,npF] (the
 * structurally non-zero columns), s[ns], C[ns,nx], D[ns,nu], G[ns,np + np_node] (row-major, compact parameter columns),
 * q[*nq] = the cone indicators of X.  Any output may be NULL.  No device needed: a maintainer checks a compiled model
 * against the closures it replaces with this call, and so do the CPU tests.
 */
int scp_model_eval_host(int model_id, const double *model_par, int N, int k, const double *x, const double *u,
                        const double *p, double *f, double *A, double *B, double *F, double *s, double *C, double *D,
                        double *G, double *q, int *nq);

int scp_problem_create(const scp_problem_desc *desc, scp_handle *out);
int scp_problem_destroy(scp_handle h);
int scp_sync(scp_handle h);
const char *scp_last_error(scp_handle h);
/*
 * Priority of the handle's HIP stream: level 0 = default, > 0 = higher, < 0 = lower (clamped to the device's range).  A caller that
 * splits a batch over several handles (sub-batches on their own streams) gives them DIFFERENT priorities so that the sub-batches do
 * not run in lockstep: with equal priorities the concurrent K3 launches share the chip evenly, finish together, and the short
 * kernels between two K3 launches (extract, discretize!, assemble) of all sub-batches then run on a nearly idle chip; with a
 * priority order the high-priority sub-batch's short kernels are dispatched ahead of the other's pending K3 workgroups and the chip
 * stays full.  The stream is drained and re-created; call it before a run is started.
 */
int scp_set_stream_priority(scp_handle h, int level);

/*
 * discretize!(ref, pbm) for a batch of B reference trajectories
 * (src/solvers/discretization.jl:160-217; FOH).  Reads
 *   xd[nx,N,B], ud[nu,N,B], p[np,B]
 * writes ref.dyn and the defects:
 *   A[nx,nx,N-1,B], Bm/Bp[nx,nu,N-1,B] (dyn.B[1], dyn.B[2]), F[nx,npF,N-1,B]
 *   (only the npF structurally non-zero columns, in Fcols order),
 *   r[nx,N-1,B], E[nx,nx,N-1,B], defect[nx,N-1,B], feas[B] (ref.feas),
 *   *seconds = device time of the call (ref.dyn.timing, :162,:214).
 * Any output pointer may be NULL to skip the copy-out (host variant).
 */
int scp_discretize_batch_host(scp_handle h, int B, const double *xd, const double *ud, const double *p,
                              double *A, double *Bm, double *Bp, double *F, double *r, double *E,
                              double *defect, uint8_t *feas, double *seconds);

/*
 * Arithmetic of discretize! on this handle: bits = 64 (default; the reference's Float64, src/utils/basic_types.jl:31)
 * or 32 -- the "fp64 vs fp32 tolerance check" of the Starship configuration: state, Phi, the cooperative LU of
 * Phi \ [..] (discretization.jl:267), the RK4 accumulators and the model evaluation all run in fp32 inside K1; inputs
 * and outputs stay fp64 arrays.  SCP_ERR_UNSUPPORTED for IMPULSE and for models without an fp32 evaluation.
 */
int scp_set_discretize_precision(scp_handle h, int bits);

/* Same on device pointers owned by the caller; asynchronous on the handle's stream.
 * feas is int32[B] on the device. */
int scp_discretize_batch_dev(scp_handle h, int B, const double *xd, const double *ud, const double *p,
                             double *A, double *Bm, double *Bp, double *F, double *r, double *E,
                             double *defect, int32_t *feas);

/*
 * propagate(sol, pbm; res) for a batch (src/solvers/discretization.jl:515-562).  FOH (:536-541): integrates the nonlinear
 * dynamics from xd[:,1,b] over LinRange(0,1,res) with the inputs linearly interpolated between the nodes and returns the
 * continuous-time state samples xc[nx,res,B] (the values of the reference's `Trajectory(tc, xc_vals, :linear)`).
 * IMPULSE (:542-560, handles created with SCP_IMPULSE): every interval restarts from its node with the model's impulse
 * response applied, x0 = xd[:,k] + f(t_k, -k, xd[:,k], ud[:,k], p), and coasts with idle inputs over LinRange(t_k, t_{k+1},
 * subres), subres = ceil(res / (N - 1)); xc[nx, 1 + (N-1) subres, B]: sample 0 = xd[:,1], then the intervals' samples (the
 * reference's sample times are those grids with the first time of every interval shifted by sqrt(eps)).  Host pointers.
 */
int scp_propagate_batch_host(scp_handle h, int B, const double *xd, const double *ud, const double *p, int res,
                             double *xc);

/* ------------------------------------------------------------------------ */
/* PTR: solve_subproblem! and the outer loop                                  */
/* ------------------------------------------------------------------------ */

/* PTR.Parameters (src/solvers/ptr.jl:57-71) minus N/Nsub/disc_method/feas_tol
 * (fixed at scp_problem_create) and `solver`/`solver_opts`, which select ECOS
 * in the reference and are replaced by the in-house structured interior-point
 * solver's options. */
typedef struct {
    int iter_max;          /* pars.iter_max                                     */
    double wvc, wtr;       /* virtual-control / trust-region weights            */
    double eps_abs, eps_rel; /* pars.ε_abs, pars.ε_rel                          */
    double q_tr, q_exit;   /* only Inf is implemented (all reference tests)     */
    /* subproblem solver options (ECOS defaults: feastol=abstol=reltol=1e-8, maxit=100) */
    int ipm_max_iter;
    double ipm_feastol, ipm_abstol, ipm_reltol;
    double ipm_reg;        /* static dual regularisation (ECOS: delta); a factorisation that breaks down is repeated with 10x */
    int ipm_nref;          /* iterative-refinement steps per Newton solve       */
    double ipm_ref_gap;    /* ... applied only once relgap < ipm_ref_gap        */
    double ipm_ref_tol;    /* ... and skipped when the residual of the computed  */
                           /* direction is below ipm_ref_tol * ipm_feastol       */
    int ipm_stall;         /* stop after this many non-improving iterations     */
    int ipm_split_step;    /* != 0: separate primal / dual step lengths when the  */
                           /* subproblem has no quadratic cost term (default 0:   */
                           /* -10 % iterations but more iteration-limit exits)   */
    /* Warm start of the subproblem solver inside scp_ptr_iterate (PTR iteration >= 2).  Every solve leaves two SNAPSHOTS in the
     * workspace: the iterates at which its complementarity measure mu = gap / degree first fell below ipm_warm_mu_coarse and
     * below ipm_warm_mu (interior points close to the central path).  The next solve of that problem starts from the fine one
     * when the previous solution moved less than ipm_warm_dev (scaled inf-norm deviation, scp.jl:909-931), from the coarse one
     * otherwise (the coarse one only if the last COLD solve of that problem needed at least ipm_warm_min_cold iterations: it
     * pays where cold solves are slow), instead of the two-solve cold start -- provided the previous solve succeeded; a
     * warm-started solve that fails is repeated cold.  Same optimum
     * (DESIGN.md section 2.1); 0 disables.  Round 6 added two more levels (fields at the end of this struct). */
    int ipm_warm;
    double ipm_warm_mu, ipm_warm_dev;
    int ipm_warm_min_cold;
    int ipm_wpe;           /* kernel variant of the subproblem solver: 0 = chosen from this handle's batch size, 1 = one */
                           /* wave per SIMD (512 registers, batches that cannot fill the chip twice), 2 = two waves per */
                           /* SIMD (callers that run several handles concurrently pass 2: the chip is shared)           */
    double ipm_warm_mu_coarse;   /* coarse snapshot level of the warm start (<= 0: 1e-1); appended in round 4 */
    /* Round 6: FOUR snapshot levels.  A solve keeps the iterates at which mu first fell below ipm_warm_mu_coarse > ipm_warm_mu_mid >
     * ipm_warm_mu > ipm_warm_mu_vfine, and the next solve of the problem starts from the FINEST level whose deviation bound covers the
     * deviation d of the previous solution: d <= ipm_warm_dev_vfine -> very fine, d <= ipm_warm_dev -> fine, d <= ipm_warm_dev_mid -> mid,
     * otherwise coarse (a level whose snapshot was never taken falls through to the next coarser one).  The right level is a matter of
     * scale: the new problem's residual at an old iterate is of the order of d; an iterate at mu = 1e-10 is the best start when the
     * reference moved by 1e-8 (the solve then needs 1 ... 3 iterations) and a trap when it moved by 1e-4 (steps of 0.01 until the
     * iteration limit of a warm attempt, then the cold repeat).  <= 0: the defaults 1e-5, 1e-1, 1e-10, 1e-6; with them ipm_warm_mu = 1e-8.
     * Two fixed rules keep the levels honest: a level is refreshed only by an iterate within two decades below it (a warm solve that
     * starts far below a level leaves that snapshot alone), and after a solve that ended ALMOST_OPTIMAL the very fine level is not used. */
    double ipm_warm_mu_mid, ipm_warm_dev_mid, ipm_warm_mu_vfine, ipm_warm_dev_vfine;
} scp_ptr_params;

/* per-problem subproblem solver exit status (MOI.TerminationStatusCode subset) */
typedef enum { SCP_SOLVER_OPTIMAL = 0, SCP_SOLVER_ALMOST_OPTIMAL = 1, SCP_SOLVER_ITERATION_LIMIT = 2,
               SCP_SOLVER_NUMERICAL_ERROR = 3 } scp_solver_status;

/* width of one history record (doubles): J, J_tr, J_vc, J_aug, deviation, improv_rel, feas,
 * solver status, solver iterations, active, gap, pres, dres, (3 reserved) */
#define SCP_HIST_WIDTH 16

/*
 * Start a batched PTR solve (PTR.solve, src/solvers/ptr.jl:448-466): uploads the initial
 * guesses xd[nx,N,B], ud[nu,N,B], p[np,B] (traj.guess) and the per-problem data
 * pp[npp,B] (Monte-Carlo initial/terminal conditions), then discretises the guess
 * (generate_initial_guess -> SubproblemSolution -> discretize!, ptr.jl:548-555).
 */
int scp_ptr_init_host(scp_handle h, int B, const scp_ptr_params *pars, const double *xd, const double *ud,
                      const double *p, const double *pp);

/*
 * Same, with the initial guesses generated ON THE DEVICE by the model's own guess rule (traj.guess,
 * src/parser/problem.jl:686-700: straight-line state, nominal input and parameter -- see csrc/models): only
 * pp[npp,B] crosses PCIe, a Monte-Carlo batch never touches the host.
 */
int scp_ptr_init_guess_host(scp_handle h, int B, const scp_ptr_params *pars, const double *pp);

/*
 * One PTR iteration for every still-active problem (ptr.jl:468-523): formulate (K2) ->
 * solve_subproblem! (K3, scp.jl:942-950) -> extract + discretize! of the new point (K1) ->
 * stopping criterion / reference update (K4).  *n_active = problems that continue; the caller
 * all-reduces it across GPUs (the only collective on the path).
 */
int scp_ptr_iterate(scp_handle h, int *n_active);

/*
 * The same iteration split in two: scp_ptr_iterate_async only ENQUEUES it on the handle's stream (several iterations may
 * be in flight; problems that have stopped are skipped on the device), scp_ptr_poll waits for the stream and returns the
 * active count of the last enqueued iteration.  A caller that splits its batch over several handles (one stream each)
 * and enqueues ahead keeps the GPU full while the slowest problems of one launch finish (the subproblem solver's
 * iteration count varies 3x between problems); scp_ptr_iterate == async + poll.
 * scp_ptr_poll_iteration returns the active count at the end of iteration `iteration` (1-based, already enqueued; 0 beyond
 * iter_max) and waits for THAT iteration only: a multi-GPU caller enqueues window k + 1, then reads and all-reduces the count of
 * window k, so the stream never drains at a window boundary (scptoolbox.jl_amd/ptr.py::group_run_resident(pipelined=True)).
 */
int scp_ptr_iterate_async(scp_handle h);
int scp_ptr_poll(scp_handle h, int *n_active);
int scp_ptr_poll_iteration(scp_handle h, int iteration, int *n_active);

/*
 * Results of the batch: final discrete trajectories (SCPSolution.xd/ud/p, scp.jl:105-119),
 * status[B] (0 = "SCP_SOLVED", 1 = "SCP_FAILED"), iterations[B], cost[4,B] = (J, J_tr, J_vc,
 * J_aug) of the last subproblem, feas[B], defect[nx,N-1,B], and the per-iteration history
 * hist[SCP_HIST_WIDTH, B, iter_max].  Any pointer may be NULL.
 */
int scp_ptr_get_host(scp_handle h, double *xd, double *ud, double *p, int32_t *status, int32_t *iterations,
                     double *cost, uint8_t *feas, double *defect, double *hist);

/* Convenience: init + iterate until no problem is active + get (single GPU). */
int scp_ptr_solve_batch_host(scp_handle h, int B, const scp_ptr_params *pars, const double *xd, const double *ud,
                             const double *p, const double *pp, double *xd_out, double *ud_out, double *p_out,
                             int32_t *status, int32_t *iterations, double *cost, uint8_t *feas, double *seconds);

/*
 * solve_subproblem!(spbm, constructor) for a batch (scp.jl:942-950 with the PTR subproblem of
 * ptr.jl:213-293): given reference trajectories, formulates and solves the convex subproblem about
 * them and returns the un-scaled solution x[nx,N,B], u[nu,N,B], p[np,B], the cost split
 * cost[4,B] = (J, J_tr, J_vc, J_aug), trust-region radii eta[2N+1,B] = (ηx, ηu, ηp), the solver
 * status / iteration count and info[8,B] = (pcost, dcost, gap, pres, dres, relgap, merit, best_it).
 * Like the reference it then discretises the new point (defect[nx,N-1,B], feas[B]).
 */
int scp_ptr_solve_subproblem_batch_host(scp_handle h, int B, const scp_ptr_params *pars, const double *xd_ref,
                                        const double *ud_ref, const double *p_ref, const double *pp, double *x,
                                        double *u, double *p, double *cost, double *eta, int32_t *solver_status,
                                        int32_t *solver_iters, double *info, double *defect, uint8_t *feas,
                                        double *seconds);

/*
 * Virtual controls and penalty epigraph variables of the LAST solved subproblem (after
 * scp_ptr_solve_subproblem_batch_host or scp_ptr_iterate) -- what the reference's SubproblemSolution(spbm) reads with
 * value(...) (src/solvers/ptr.jl:399-432): vd[nx,N-1,B] (E_k vd_k = linearised dynamics defect, with the reference's
 * discretised E_k = ref.dyn.E, ptr.jl:805), vs[ns,N,B], vic[nic,B], vtc[ntc,B], P[N,B] (P_k = ||E_k vd_k||_1 +
 * ||vs_k||_1, ptr.jl:813-887), Pf[2,B] = (||vic||_1, ||vtc||_1).  They are eliminated analytically from the reduced
 * subproblem the device solves and re-evaluated from its solution.  Any pointer may be NULL.
 */
int scp_ptr_get_virtual_controls_host(scp_handle h, double *vd, double *vs, double *vic, double *vtc, double *P,
                                      double *Pf);

/*
 * `traj.guess(N)` (src/parser/problem.jl:686-700) of the compiled model for a Monte-Carlo batch, evaluated on the device
 * (§8(f)4): pp[npp,B] -> xd[nx,N,B], ud[nu,N,B], p[np,B] on the host.  Any registered model: straight-line guesses
 * (quadrotor/definition.jl:60-90 and the builder-defined problems), the free-flyer's axis-by-axis path with SLERP attitude
 * (freeflyer/definition.jl:84-186, quaternion.jl:483-490), and -- since round 4 -- the Starship's own guess
 * (starship_flip/definition.jl:97-445) for every instance: bang-bang flip simulated with RK4 (one thread per instance), the
 * terminal-descent programs of all candidate durations as one batch of the conic engine, reconstruction of attitude / thrust /
 * rate / mass on the device.  An instance for which the reference would raise an error (no velocity crossing, no feasible descent
 * duration) gets the straight-line guess instead; scp_guess_failures returns how many of the last call did.
 */
int scp_guess_batch_host(scp_handle h, int B, const double *pp, double *xd, double *ud, double *p);
int scp_guess_failures(scp_handle h);

/* Restart the batch from the initial guesses uploaded by the last scp_ptr_init_host, entirely on the
 * device (D2D copy + discretize! of the guess): the inputs stay resident in HBM. */
int scp_ptr_restart(scp_handle h);

/* Cumulative device seconds (HIP events on the handle's stream) and launch counts per kernel since the
 * last reset: index 0 discretize (K1), 1 assemble (K2), 2 structured IPM (K3), 3 extract+update (K4).
 * reset != 0 clears the counters after reading. */
int scp_get_kernel_timing(scp_handle h, double seconds[4], long launches[4], int reset);

/* Diagnostic: phase counters of the last structured-IPM launch for problem b (100 MHz wall-clock ticks):
 * G, G', factor, rhs+forward, backward, aux recovery, -, total. */
int scp_debug_get_ipm_profile(scp_handle h, int b, long long *ticks8);

/* Diagnostic: copy out the assembled stage-form subproblem data of problem b (layout of
 * csrc/stage_problem.hpp) after scp_ptr_solve_subproblem_batch_host / scp_ptr_iterate.
 * *n_doubles returns the slab length; buf may be NULL to query it. */
int scp_debug_get_stage_problem(scp_handle h, int b, double *buf, long *n_doubles);

/* ------------------------------------------------------------------------ */
/* Generic subproblem pipeline: any SCP subproblem as a conic template        */
/* (PTR with q_tr in {1,2,4,Inf}, SCvx, correct_convex!, ...)                 */
/* ------------------------------------------------------------------------ */

/*
 * The reference formulates a new JuMP model per iteration (`Subproblem(pbm, iter, ref)` + add_dynamics! ... add_cost!,
 * src/solvers/ptr.jl:213-293,467-480; scvx.jl:225-303,483-490; scp.jl:657-895).  Every coefficient it writes is affine
 * in a few per-problem numbers -- ref.dyn, the reference trajectory, the Jacobians of s and of the boundary conditions
 * about it, the trust-region radius.  The host formulates ONCE (scptoolbox.jl_amd/subproblem.py) and hands over
 *   - the sparsity pattern of the standard form of include/scp_conic.h, and
 *   - for each value array an affine map  value[i] = val0[i] + sum_{t = ptr[i]}^{ptr[i+1]-1} coef[t] * src[sidx[t]]
 * over the per-problem SOURCE VECTOR the device fills (layout: scp_sub_source_layout).
 */
typedef struct {
    int len;
    const double *val0; /* [len]                */
    const int *ptr;     /* [len + 1]            */
    const int *sidx;    /* [ptr[len]]           */
    const double *coef; /* [ptr[len]]           */
} scp_affine_map;

typedef struct {
    int n, p, m, l, ncones;     /* standard form sizes, m = l + sum(q)                                   */
    const int *q;
    const int *Pp, *Pi, *Ap, *Ai, *Gp, *Gi; /* CSC patterns (P: upper triangle)                           */
    scp_affine_map c, b, h, Gx, Ax, Px;     /* value maps over the source vector                          */
    int nsrc, nscal;            /* source vector length / trailing algorithm scalars (SCvx: eta)          */
    const int *ix, *iu, *ip;    /* positions of the scaled variables xh[nx,N], uh[nu,N], ph[np] in x      */
    int nfun;                   /* linear functionals of the conic solution reported per problem           */
    scp_affine_map fun;         /* fun[j] = val0[j] + sum coef * x[sidx]  (e.g. trapz(P) + sum(Pf))        */
} scp_sub_template;

typedef struct scp_sub *scp_sub_handle;

/* offsets[0..20] of the 20 source segments (doubles, per problem; every segment column-major):
 * xref(nx,N) uref(nu,N) pref(np + np_node N) A(nx,nx,N-1) Bm Bp(nx,nu,N-1) F(nx,npF,N-1) r(nx,N-1) E(nx,nx,N-1) C(ns,nx,N)
 * D(ns,nu,N) Gs(ns,np + np_node,N) rs(ns,N) H0(nic,nx) K0(nic,np) l0(nic) Hf(ntc,nx) Kf(ntc,np) lf(ntc) scal(nscal);
 * *nsrc = total.  Gs is the COMPACT parameter Jacobian of s at its node (scp_model_info). */
int scp_sub_source_layout(scp_handle h, int nscal, int *offsets, int *nsrc);

int scp_sub_create(scp_handle h, const scp_sub_template *T, scp_sub_handle *out);
int scp_sub_destroy(scp_sub_handle s);
/* statistics of the subproblem's conic engine: the 16 values of scp_conic_stats (include/scp_conic.h) */
int scp_sub_stats(scp_sub_handle s, long long stats[16]);
const char *scp_sub_last_error(scp_sub_handle s);

/*
 * solve_subproblem!(spbm, constructor) (src/solvers/scp.jl:942-950) for a batch and ANY template: discretises the
 * reference trajectories xd_ref[nx,N,B], ud_ref[nu,N,B], p_ref[np,B] (discretize!), linearises the non-convex
 * constraints / boundary conditions about them, fills the conic values (gather), solves (conic_ipm_kernel), un-scales
 * x[nx,N,B], u[nu,N,B], p[np,B] (value(blk), block.jl:368-394) and discretises the new point (defect[nx,N-1,B],
 * feas[B]).  scal[nscal,B]: algorithm scalars; fun[nfun,B]; xconic[n,B]: the whole conic solution (virtual controls,
 * epigraph variables, ...); status/iters/info[8,B] as scp_conic_solve_batch_host.  Any output may be NULL.
 */
int scp_sub_solve_batch_host(scp_sub_handle s, int B, const double *xd_ref, const double *ud_ref, const double *p_ref,
                             const double *pp, const double *scal, const scp_conic_opts *opts, double *x, double *u,
                             double *p, double *fun, double *xconic, int32_t *status, int32_t *iters, double *info,
                             double *defect, uint8_t *feas, double *seconds);

/* SCvx.Parameters (src/solvers/scvx.jl:60-81) minus N/Nsub/disc_method/feas_tol (fixed at scp_problem_create) and
 * q_tr (fixed by the template). */
typedef struct {
    int iter_max;
    double lam;                       /* λ: virtual-control penalty weight                  */
    double rho_0, rho_1, rho_2;       /* ρ thresholds of the update rule (scvx.jl:1000-1045) */
    double beta_sh, beta_gr;          /* shrink / growth factors                             */
    double eta_init, eta_lb, eta_ub;  /* trust-region radius                                 */
    double eps_abs, eps_rel;
    double q_exit;                    /* norm of solution_deviation (scp.jl:909-931): any q >= 1 or Inf */
    scp_conic_opts solver;            /* subproblem solver options (pars.solver_opts)        */
} scp_scvx_params;

/* width of one SCvx history record: L, L_pen, L_aug, J_ref, J_sol, pre_improv, act_improv, rho, eta, eta_next, accepted,
 * stop, deviation, feas, solver status, solver iterations */
#define SCP_SCVX_HIST_WIDTH 16

/*
 * SCvx.solve (src/solvers/scvx.jl:459-540) for a batch, resident on the device.  `sub`: an SCvx subproblem template
 * (nscal = 1: eta; fun[0] = trapz(P) + sum(Pf)); `proj`: a correct_convex! template of the same problem handle or NULL
 * (generate_initial_guess projects the guess, scvx.jl:555-565, scp.jl:275-361).  init uploads the guesses and pp[npp,B];
 * iterate = formulate + solve_subproblem! + discretize! + check_stopping_criterion! + update_trust_region!
 * (scvx.jl:711-770, 924-1045) for every active problem; get returns the LAST subproblem solution of every problem
 * (SCPSolution(history), scp.jl:196-245), status[B] (0 solved, 1 failed, 2 guess projection failed), iterations[B],
 * cost[2,B] = (J of the reference, J of the last solution), feas[B], defect, hist[SCP_SCVX_HIST_WIDTH, B, iter_max].
 */
int scp_scvx_init_host(scp_sub_handle sub, scp_sub_handle proj, int B, const scp_scvx_params *pars, const double *xd,
                       const double *ud, const double *p, const double *pp);
int scp_scvx_iterate(scp_sub_handle sub, int *n_active);
int scp_scvx_get_host(scp_sub_handle sub, double *xd, double *ud, double *p, int32_t *status, int32_t *iterations,
                      double *cost, uint8_t *feas, double *defect, double *hist);

/* GuSTO.Parameters (src/solvers/gusto.jl:59-85) minus N/Nsub/disc_method/feas_tol (fixed at scp_problem_create).  pen / hom must
 * be those the template was built with (:softplus = exponential cones in the template).  q_tr must be the norm the template was built with: the update rule
 * measures the trust-region violation of the new point in it (gusto.jl:1172-1185, 1318-1340). */
typedef struct {
    int iter_max;
    double lam_init, lam_max;         /* soft-penalty weight: initial value, failure threshold                     */
    double rho_0, rho_1;              /* model-accuracy thresholds of the update rule (gusto.jl:1310-1427)          */
    double beta_sh, beta_gr;          /* trust-region shrink / growth factors                                       */
    double gamma_fail;                /* lambda growth factor after an infeasible / trust-violating step            */
    double eta_init, eta_lb, eta_ub;  /* trust-region radius                                                        */
    double mu;                        /* eta *= mu^(1 + k - iter_mu) for k >= iter_mu (kappa, gusto.jl:264)          */
    int iter_mu;
    double eps_abs, eps_rel;
    double q_tr, q_exit;              /* trust-region norm / norm of solution_deviation: q >= 1 or Inf                */
    int nst;                          /* soft-penalised quantities per node in the template: the cone indicators of X
                                         (scp_model_state_indicators) + ns; anything else is refused                  */
    scp_conic_opts solver;
    int pen;                          /* soft penalty (gusto.jl:79-80, 966-1031): 0 = :quad, lambda max(0, f)^2 (default); 1 = :softplus,
                                         lambda log(1 + exp(hom f)) / hom through exponential cones (the template's penalty
                                         variables are then the w of gusto.jl:1000-1029).  Appended in round 4. */
    double hom;                       /* homotopy parameter of :softplus                                              */
} scp_gusto_params;

/* columns of one GuSTO history record (width SCP_SCVX_HIST_WIDTH): L, L_st, L_tr, J_aug, J_st, rho, eta, lambda, eta_next,
 * lambda_next, flags (1 accepted | 2 stop | 4 trust region violated | 8 constraints feasible | 16 dynamically feasible),
 * deviation, solver status, solver iterations, dynamics error, its normalisation */

/*
 * GuSTO.solve (src/solvers/gusto.jl:425-502) for a batch, resident on the device.  `sub`: a GuSTO template (nscal = 2:
 * eta, lambda -- lambda weights the QUADRATIC cost, i.e. the P values are per problem; fun = the penalty variables
 * v_tr[N] then v_st[nst, N]); `proj`: a correct_convex! template or NULL (generate_initial_guess, gusto.jl:516-521).
 * iterate = formulate + solve_subproblem! + discretize! + solution costs (:391-407) + check_stopping_criterion!
 * (:1203-1230) + update_trust_region! (:1245-1293, 1310-1427).  get: as scp_scvx_get_host with cost[2,B] = (J_aug of the
 * reference, J_aug of the last solution) and the GuSTO history columns.
 */
int scp_gusto_init_host(scp_sub_handle sub, scp_sub_handle proj, int B, const scp_gusto_params *pars, const double *xd,
                        const double *ud, const double *p, const double *pp);
int scp_gusto_iterate(scp_sub_handle sub, int *n_active);
int scp_gusto_get_host(scp_sub_handle sub, double *xd, double *ud, double *p, int32_t *status, int32_t *iterations,
                       double *cost, uint8_t *feas, double *defect, double *hist);

/* PTR.Parameters (src/solvers/ptr.jl:57-71) for the generic path: minus N/Nsub/disc_method/feas_tol (fixed at scp_problem_create)
 * and q_tr (fixed by the template); cost_const = the constant the template's objective omits (cost of the scaling offsets). */
typedef struct {
    int iter_max;
    double wvc, wtr;
    double eps_abs, eps_rel;
    double q_exit;                    /* norm of solution_deviation (scp.jl:909-931): any q >= 1 or Inf */
    double cost_const;
    scp_conic_opts solver;
} scp_ptr_generic_params;

/*
 * PTR.solve (src/solvers/ptr.jl:448-532) for a batch, resident on the device, over ANY PTR template (build_ptr: q_tr in
 * {1, 2, 4, Inf}; fun[0] = trapz(P) + sum(Pf), fun[1] = trapz(eta_x) + trapz(eta_u) + eta_p) -- the loop of the models
 * without the stage-structured fast path (Starship, free-flyer) and of the non-Inf trust-region norms.  iterate = formulate +
 * solve_subproblem! + discretize! + cost split (:753-895) + check_stopping_criterion! (:908-932) + ref = sol (:509) for
 * every active problem.  get: as scp_ptr_get_host (cost[4,B] = J, J_tr, J_vc, J_aug of the last subproblem; hist with the
 * SCP_HIST_WIDTH columns).
 */
int scp_ptr_generic_init_host(scp_sub_handle sub, int B, const scp_ptr_generic_params *pars, const double *xd, const double *ud,
                              const double *p, const double *pp);
int scp_ptr_generic_iterate(scp_sub_handle sub, int *n_active);
int scp_ptr_generic_get_host(scp_sub_handle sub, double *xd, double *ud, double *p, int32_t *status, int32_t *iterations,
                             double *cost, uint8_t *feas, double *defect, double *hist);

/* ---------------------------------------------------------------------------------------------------------------------------
 * Multi-GPU behind the boundary (reference: the sequential Monte-Carlo loop `for trial = 1:num_trials`,
 * test/examples/quadrotor/tests.jl:171-184, is the axis that is sharded).  One process per GPU; every rank owns a contiguous
 * shard of the batch (scp_shard_range) in one or more handles (sub-batches on their own HIP streams); the ONLY collective is the
 * per-window all-reduce (SUM, one int64) of the number of still-active problems, issued by RCCL ON THE DEVICE: the count never
 * visits the host between the update kernel that produces it and the collective that sums it.
 *
 *   rank 0:  scp_comm_unique_id(id)  ->  ship the 128 bytes to the other ranks (MPI.bcast, a file, torch.distributed ...)
 *   all   :  scp_comm_create(id, rank, world, device, &comm)            (ncclCommInitRank; librccl.so is loaded at this call)
 *            scp_ptr_init_guess_host(part[i], B_i, &pars, pp_i)         (this rank's shard, as on one GPU)
 *            scp_ptr_run_sharded(comm, part, nparts, lookahead, &iterations, &collectives)
 *            scp_ptr_get_host(part[i], ...)
 *
 * scp_ptr_run_sharded enqueues WINDOWS of `lookahead` PTR iterations on every handle's stream; behind each window a one-thread
 * kernel sums the handles' device-resident active counts and ncclAllReduce adds the ranks' sums on a separate high-priority
 * stream (ordered by events, no stream is drained), the result lands in a pinned ring.  The host enqueues window w + 1 BEFORE it
 * reads the global count of window w, so neither the compute streams nor the host wait for a collective; every rank reads the
 * same sequence of global counts, hence all ranks enqueue the same number of windows (lockstep) -- one more than needed, whose
 * launches skip the stopped problems on the device.  comm == NULL (or world == 1): the same loop without RCCL.
 * *iterations = PTR iterations executed until no problem was active on any rank (<= iter_max), *collectives = all-reduces issued.
 * ------------------------------------------------------------------------------------------------------------------------- */
#define SCP_COMM_ID_BYTES 128
typedef struct scp_comm *scp_comm_handle;
int scp_comm_unique_id(unsigned char id[SCP_COMM_ID_BYTES]);
/* the LOCAL steps scp_comm_create takes before its collective part (load RCCL, select the device, create a stream), without the collective:
 * a host calls it on every rank and agrees on the results (MPI.Allreduce, torch.distributed ...) BEFORE scp_comm_create, whose
 * ncclCommInitRank would otherwise wait for ever for a rank that failed locally; the error text is in scp_comm_last_error(NULL) */
int scp_comm_preflight(int device);
int scp_comm_create(const unsigned char id[SCP_COMM_ID_BYTES], int rank, int world, int device, scp_comm_handle *out);
void scp_comm_destroy(scp_comm_handle c);
const char *scp_comm_last_error(scp_comm_handle c);
/* blocking helper on the communicator's stream (tests, final reductions of scalar statistics): *value <- SUM over ranks */
int scp_comm_all_reduce_sum_i64(scp_comm_handle c, long long *value);
/* contiguous shard [lo, hi) of a global batch of n_total problems owned by `rank` of `world` (sizes differ by at most one) */
void scp_shard_range(long n_total, int rank, int world, long *lo, long *hi);
int scp_ptr_run_sharded(scp_comm_handle c, scp_handle *parts, int nparts, int lookahead, int *iterations, int *collectives);

#ifdef __cplusplus
}
#endif
#endif /* SCP_MI355X_H */
