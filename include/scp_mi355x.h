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
    SCP_ERR_UNSUPPORTED = 7
} scp_status;

/* model registry (replaces traj.f/A/B/F..., src/parser/problem.jl:432-450) */
typedef enum {
    SCP_MODEL_DOUBLE_INTEGRATOR = 0, /* builder-defined, see DESIGN.md       */
    SCP_MODEL_QUADROTOR = 1,         /* test/examples/quadrotor              */
    SCP_MODEL_ROCKET_LANDING = 2     /* builder-defined over rocket_landing  */
} scp_model_id;

/* DiscretizationType, src/parser/problem.jl:52 */
typedef enum { SCP_FOH = 0, SCP_IMPULSE = 1 } scp_disc_method;

/* Static description of a model (dimensions the caller needs to size buffers). */
typedef struct {
    int nx, nu, np;   /* state / input / parameter dims (problem_set_dims!)     */
    int npF;          /* number of structurally non-zero columns of F (F8)      */
    int Fcols[8];     /* their 0-based column indices into p                    */
    int ns;           /* rows of the non-convex path constraint s               */
    int nic, ntc;     /* rows of the initial / terminal boundary conditions     */
    int npar;         /* doubles in the shared model parameter blob             */
    int npp;          /* doubles of per-problem data (Monte-Carlo ICs)          */
} scp_model_info;

/* SCPScaling, src/solvers/scp.jl:39-49 (diagonals only; the reference's
 * Sx/Su/Sp are diagonal matrices, scp.jl:489-511). */
typedef struct {
    const double *Sx, *cx; /* [nx] */
    const double *Su, *cu; /* [nu] */
    const double *Sp, *cp; /* [np] */
} scp_scaling;

typedef struct {
    int model_id;            /* scp_model_id                                    */
    const double *model_par; /* [npar] shared model parameters                   */
    int N;                   /* temporal grid nodes (pars.N)                     */
    int Nsub;                /* sub-interval integration nodes (pars.Nsub)       */
    int disc_method;         /* scp_disc_method (pars.disc_method)               */
    double feas_tol;         /* dynamic feasibility tolerance (pars.feas_tol)    */
    scp_scaling scale;       /* variable scaling (pbm.common.scale)              */
    int batch_capacity;      /* max problems per call                            */
    int device;              /* HIP device ordinal                               */
} scp_problem_desc;

int scp_model_query(int model_id, scp_model_info *info);

int scp_problem_create(const scp_problem_desc *desc, scp_handle *out);
int scp_problem_destroy(scp_handle h);
int scp_sync(scp_handle h);
const char *scp_last_error(scp_handle h);

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

/* Same on device pointers owned by the caller; asynchronous on the handle's stream.
 * feas is int32[B] on the device. */
int scp_discretize_batch_dev(scp_handle h, int B, const double *xd, const double *ud, const double *p,
                             double *A, double *Bm, double *Bp, double *F, double *r, double *E,
                             double *defect, int32_t *feas);

#ifdef __cplusplus
}
#endif
#endif /* SCP_MI355X_H */
