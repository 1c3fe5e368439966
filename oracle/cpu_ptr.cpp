// CPU BASELINE (bench.py `cpu_baseline` leg and tests only -- NOT product code, never loaded by the product path).
//
// A plain C++ / OpenMP restatement of the SAME PTR iteration the HIP library runs, so that the GPU throughput can be
// quoted beside an honest all-cores CPU number (the reference's Julia + ECOS path cannot run in this
// image -- no Julia):
//   discretize!        oracle/scp_oracle.c (C restatement of src/solvers/discretization.jl:160-406), linked in
//   formulate          the product's stage-form assembly ptr_assemble_entry<M> (csrc/stage_problem.hpp, __host__
//                      __device__) and model definitions (csrc/models/*.hpp) compiled for the host: both solvers see
//                      bit-identical subproblem data
//   solve_subproblem!  the structured primal-dual interior-point method of oracle/ipm_struct.py / csrc/ipm2_*.hpp
//                      (Mehrotra predictor-corrector, NT scaling, quasi-definite block sweep + arrow column, static
//                      regularisation + one refinement step) in scalar C++ with small dense blocks
//   outer loop         src/solvers/ptr.jl:448-532 (fixed iteration count, reference update) per problem
// One problem per OpenMP task (`omp parallel for` over the batch), single-threaded inside a problem like the reference.
// Build: hipcc -x hip (host pass only; no kernel is instantiated), see oracle/Makefile.
#include <hip/hip_runtime.h>
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../scptoolbox.jl_amd/csrc/stage_problem.hpp"
#include "../scptoolbox.jl_amd/csrc/models/double_integrator.hpp"
#include "../scptoolbox.jl_amd/csrc/models/quadrotor.hpp"
#include "../scptoolbox.jl_amd/csrc/models/rocket_landing.hpp"

extern "C" int oracle_discretize(int model_id, const double* par, int N, int Nsub, const double* xd, const double* ud,
                                 const double* p, const double* iSx_diag, double feas_tol, double* A, double* Bm,
                                 double* Bp, double* F, double* r, double* E, double* defect, int* feas);

namespace {
using namespace scp;

struct IpmOpts {
    int max_iter = 100, nref = 1, stall = 3;
    double feastol = 1e-8, abstol = 1e-8, reltol = 1e-8, reg = 1e-12, ref_gap = 1e-2;
    // warm start of the IPM (environment SCP_CPU_WARM=mode, SCP_CPU_WARM_MU, SCP_CPU_WARM_FROM, ...): 0 cold (two-solve ECOS-style
    // point), 4 = what the device does since round 4: two snapshots of the previous solve (the iterates where mu first fell
    // below warm_save_mu_coarse / warm_save_mu), the fine one when the reference moved less than warm_dev, else the coarse one.
    // Experiments kept for the record: 1 structured centred point about the reference, 2 previous FINAL iterate pushed into the
    // interior (the device's round-2/3 scheme: ~32 iterations per warm solve against ~15), 3 fine snapshot only.
    int warm = 5, warm_from = 1, warm_min_cold = 25, warm_max_iter = 45;
    // warm == 5 (round 6, what the device does now): NL snapshot levels -- the iterates at which mu first fell below lvl_mu[l] -- and the
    // next solve starts from the FINEST level l whose deviation bound covers the previous solution's deviation, prev_dev <= lvl_dev[l]
    // (and whose snapshot exists; level 0 only where cold solves are slow, warm_min_cold).  A snapshot at mu = 1e-9 is the right start when
    // the reference moved by 1e-8 and the wrong one when it moved by 1e-4: the new problem's residual at that point (~ the deviation)
    // is then far above the distance to the boundary and the iteration crawls with steps of 0.01 (45 iterations, then the cold repeat).
    static constexpr int NL = 4;
    double lvl_mu[NL] = {1e-1, 1e-5, 1e-8, 1e-10};
    double lvl_dev[NL] = {1e300, 1e-1, 1e-3, 1e-6};
    int lvl_cap[NL] = {45, 45, 45, 16};   // iteration limit of a warm attempt from level l (experiment: SCP_CPU_LVL_CAP)
    int almost_lvl = NL - 2;              // finest level used after a solve that ended ALMOST_OPTIMAL (its last iterates are not well centred; SCP_CPU_ALMOST_LVL)
    double lvl_floor = 1e-2;              // > 0: an iterate refreshes level l only if lvl_mu[l] * lvl_floor <= mu <= lvl_mu[l] (SCP_CPU_LVL_FLOOR)
    int cross = 0;                        // 1: a level's snapshot is the iterate that CROSSES the level (mu_prev > level >= mu); a warm solve
                                          // that starts below a level leaves that level's snapshot alone (SCP_CPU_CROSS=0: first iterate below)
    int ref_on_stall = 0;         // experiment: refine once the merit has not improved for this many iterations
    int ref_corrector_only = 0;   // experiment: no refinement of the predictor (affine) direction
    double warm_mu = 1e-5, warm_dev = 1e-3, warm_save_mu = 1e-9, warm_save_mu_coarse = 1e-1;   // (fine level 1e-9 since round 6, like the device)
    int reg_escalate = 4;
    double stall_rel = 1.0;
    double step_frac = 0.99, cgamma = 0.0;   // experiments: SCP_CPU_STEPFRAC, SCP_CPU_CGAMMA
};
struct IpmResult {
    int status = 2, iters = 0;   // 0 OPTIMAL, 1 ALMOST_OPTIMAL, 2 ITERATION_LIMIT, 3 NUMERICAL_ERROR
    double pcost = 0, dcost = 0, gap = 0, pres = 0, dres = 0, relgap = 0;
};

// in-place lower Cholesky of the n x n row-major matrix A (leading dimension ld); false on a non-positive pivot
static bool chol(double* A, int n, int ld)
{
    for (int j = 0; j < n; j++) {
        double d = A[j * ld + j];
        for (int q = 0; q < j; q++) d -= A[j * ld + q] * A[j * ld + q];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        A[j * ld + j] = d;
        for (int i = j + 1; i < n; i++) {
            double v = A[i * ld + j];
            for (int q = 0; q < j; q++) v -= A[i * ld + q] * A[j * ld + q];
            A[i * ld + j] = v / d;
        }
    }
    return true;
}
static inline void lsolve(const double* L, int n, int ld, double* x)   // x <- L^-1 x
{
    for (int i = 0; i < n; i++) {
        double v = x[i];
        for (int q = 0; q < i; q++) v -= L[i * ld + q] * x[q];
        x[i] = v / L[i * ld + i];
    }
}
static inline void ltsolve(const double* L, int n, int ld, double* x)  // x <- L^-T x
{
    for (int i = n - 1; i >= 0; i--) {
        double v = x[i];
        for (int q = i + 1; q < n; q++) v -= L[q * ld + i] * x[q];
        x[i] = v / L[i * ld + i];
    }
}

template <class M>
struct CpuIpm {
    using S = SP<M>;
    static constexpr int nx = S::nx, nu = S::nu, np = S::np, npa = S::npa, nz = S::nz, ns = S::ns, nl = S::nl,
                         nsoc = S::nsoc, ml = S::ml, ng = S::ng, nic = S::nic, ntc = S::ntc, nbc = S::nbc, RS = S::RS,
                         RG = S::RG, AS = S::AS, AG = S::AG, MNU = S::MNU, SR = S::SR;
    static constexpr int NS1 = ns > 0 ? ns : 1, NSOC1 = nsoc > 0 ? nsoc : 1;
    int N;
    const double* P;   // slab
    typename S::Off o;
    long XI, ROWS;
    // factor storage per node
    std::vector<double> Lz, Lnu, X, Y, Dt, Ft, cf, C0, Ycz, Ycnu, socW, spL;
    std::vector<double> fb, ft;
    std::vector<double> xi_prev, lam_prev;   // final iterate of the previous solve (warm-start experiments)
    std::vector<double> xi_snap, s_snap, lam_snap;   // warm == 3: the iterate at which mu first fell below warm_save_mu (a well-centred point)
    bool snap_ok = false;
    std::vector<double> xi_snapA, s_snapA, lam_snapA;   // warm == 4: coarse snapshot (mu <= warm_save_mu_coarse) for large reference deviations
    bool snapA_ok = false; int snap_level = 1;   // level the next warm solve starts from: 0 coarse, 1 fine (warm == 5: 0 .. NL - 1)
    std::vector<double> xi_sn[IpmOpts::NL], s_sn[IpmOpts::NL], lam_sn[IpmOpts::NL];   // warm == 5
    bool sn_ok[IpmOpts::NL] = {false, false, false, false};
    bool sn_ok_prev[IpmOpts::NL] = {false, false, false, false};   // availability before the current solve
    double sn_acc[IpmOpts::NL] = {0, 0, 0, 0};   // reference deviation accumulated since the level's snapshot was taken
    bool sn_new[IpmOpts::NL] = {false, false, false, false};   // taken by the last solve
    bool use_warm = false;
    long gondzio_tried = 0, gondzio_kept = 0;
    IpmOpts opt;

    // ---- slab views ----
    const double* st(int k) const { return P + (long)k * SR; }
    const double* Dm(int k) const { return st(k) + S::O_D; }
    const double* Em(int k) const { return st(k) + S::O_E; }
    const double* Fp(int k) const { return st(k) + S::O_FP; }
    const double* Kl(int k) const { return st(k) + S::O_KL; }
    const double* Kp(int k) const { return st(k) + S::O_KP; }
    const double* G() const { return P + o.glob; }
    // ---- vector accessors (device layout) ----
    double& Z(double* v, int k, int j) const { return v[(long)k * nz + j]; }
    double& AUX(double* v, int k, int i) const { return v[(long)N * nz + (long)k * AS + i]; }
    double& PV(double* v, int j) const { return v[(long)N * (nz + AS) + j]; }
    double& GAUX(double* v, int i) const { return v[(long)N * (nz + AS) + npa + i]; }
    double& ROW(double* v, int k, int r) const { return v[(long)k * RS + r]; }
    double& GROW(double* v, int r) const { return v[(long)N * RS + r]; }
    bool live(int k, int r) const { return !(k == N - 1 && r < 2 * nx); }
    bool is_dead(long i) const { return i >= (long)(N - 1) * RS && i < (long)(N - 1) * RS + 2 * nx; }
    bool is_soc(long i) const { return nsoc > 0 && i < (long)N * RS && (i % RS) >= S::R_SOC; }

    void bind(const double* slab, int N_)
    {
        N = N_; P = slab; o = S::offsets(N);
        XI = (long)N * (nz + AS) + npa + AG; ROWS = (long)N * RS + RG;
        Lz.assign((size_t)N * nz * nz, 0); Lnu.assign((size_t)N * MNU * MNU, 0); X.assign((size_t)N * MNU * nz, 0);
        Y.assign((size_t)N * nz * MNU, 0); Dt.assign((size_t)N * MNU * nz, 0); Ft.assign((size_t)N * MNU * npa, 0);
        cf.assign((size_t)N * MNU * 2, 0); C0.assign((size_t)N * nz * npa, 0); Ycz.assign((size_t)N * nz * npa, 0);
        Ycnu.assign((size_t)N * MNU * npa, 0); socW.assign((size_t)N * NSOC1 * 36, 0); spL.assign(npa * npa, 0);
        fb.assign((size_t)N * nz, 0); ft.assign((size_t)N * MNU, 0);
    }

    // main-variable part of row r of node k
    double row_main(const double* v, int k, int r) const
    {
        const double* zk = v + (long)k * nz;
        const double* pv = v + (long)N * (nz + AS);
        if (r < 2 * nx) {
            if (k >= N - 1) return 0.0;
            const int i = r % nx;
            const double* zn = zk + nz;
            double acc = 0.0;
            for (int j = 0; j < nz; j++) acc += Dm(k)[i * nz + j] * zk[j] + Em(k)[i * nz + j] * zn[j];
            for (int j = 0; j < np; j++) acc += Fp(k)[i * npa + j] * pv[j];
            return r < nx ? acc : -acc;
        }
        if (r < S::R_TR0) {
            if (r >= S::R_H1) return 0.0;
            const int i = r - S::R_H0;
            double acc = 0.0;
            for (int j = 0; j < nz; j++) acc += Kl(k)[i * nz + j] * zk[j];
            for (int j = 0; j < np; j++) acc += Kp(k)[i * npa + j] * pv[j];
            return acc;
        }
        if (r < S::R_LIN) { const int j = (r - S::R_TR0) % nz; return r < S::R_TR1 ? zk[j] : -zk[j]; }
        const int row = ns + (r - S::R_LIN);
        double acc = 0.0;
        for (int j = 0; j < nz; j++) acc += Kl(k)[row * nz + j] * zk[j];
        for (int j = 0; j < np; j++) acc += Kp(k)[row * npa + j] * pv[j];
        return r < S::R_SOC ? acc : -acc;
    }
    double row_aux(double* v, int k, int r) const
    {
        if (r < 2 * nx) return AUX(v, k, S::A_Y + r % nx);
        if (r < S::R_TR0) return AUX(v, k, S::A_V + (r - S::R_H0) % NS1);
        if (r < S::R_LIN) return AUX(v, k, ((r - S::R_TR0) % nz) < nx ? S::A_EX : S::A_EU);
        return 0.0;
    }
    // boundary-condition row activity (main part): which = 0 ic, 1 tc
    double bc_main(const double* v, int which, int i) const
    {
        const double* H = G() + (which == 0 ? S::Q_H0 : S::Q_HF);
        const double* K = G() + (which == 0 ? S::Q_K0 : S::Q_KF);
        const double* zk = v + (long)(which == 0 ? 0 : N - 1) * nz;
        const double* pv = v + (long)N * (nz + AS);
        double acc = 0.0;
        for (int j = 0; j < nx; j++) acc += H[i * nx + j] * zk[j];
        for (int j = 0; j < np; j++) acc += K[i * npa + j] * pv[j];
        return acc;
    }
    void G_apply(double* v, double* out) const
    {
        for (int k = 0; k < N; k++)
            for (int r = 0; r < RS; r++) ROW(out, k, r) = live(k, r) ? row_main(v, k, r) - row_aux(v, k, r) : 0.0;
        for (int i = 0; i < nic; i++) { const double a = bc_main(v, 0, i), y = GAUX(v, S::GA_YIC + i); GROW(out, S::G_IC0 + i) = a - y; GROW(out, S::G_IC1 + i) = -a - y; }
        for (int i = 0; i < ntc; i++) { const double a = bc_main(v, 1, i), y = GAUX(v, S::GA_YTC + i); GROW(out, S::G_TC0 + i) = a - y; GROW(out, S::G_TC1 + i) = -a - y; }
        for (int j = 0; j < np; j++) { GROW(out, S::G_TRP0 + j) = PV(v, j) - GAUX(v, S::GA_EP); GROW(out, S::G_TRP1 + j) = -PV(v, j) - GAUX(v, S::GA_EP); }
        for (int i = 0; i < ng; i++) {
            double acc = 0.0;
            for (int j = 0; j < np; j++) acc += G()[S::Q_LP + i * npa + j] * PV(v, j);
            GROW(out, S::G_LIN + i) = acc;
        }
    }
    void GT_apply(double* mu, double* out) const
    {
        std::fill(out, out + XI, 0.0);
        for (int k = 0; k < N; k++) {
            double* zk = out + (long)k * nz;
            double* pv = out + (long)N * (nz + AS);
            if (k < N - 1) {
                double* zn = zk + nz;
                for (int i = 0; i < nx; i++) {
                    const double d = ROW(mu, k, i) - ROW(mu, k, nx + i);
                    for (int j = 0; j < nz; j++) { zk[j] += Dm(k)[i * nz + j] * d; zn[j] += Em(k)[i * nz + j] * d; }
                    for (int j = 0; j < np; j++) pv[j] += Fp(k)[i * npa + j] * d;
                    AUX(out, k, S::A_Y + i) = -(ROW(mu, k, i) + ROW(mu, k, nx + i));
                }
            }
            for (int j = 0; j < nz; j++) {
                zk[j] += ROW(mu, k, S::R_TR0 + j) - ROW(mu, k, S::R_TR1 + j);
                AUX(out, k, j < nx ? S::A_EX : S::A_EU) -= ROW(mu, k, S::R_TR0 + j) + ROW(mu, k, S::R_TR1 + j);
            }
            for (int row = 0; row < ml; row++) {
                double m;
                if (row < ns) { m = ROW(mu, k, S::R_H0 + row); AUX(out, k, S::A_V + row) = -(ROW(mu, k, S::R_H0 + row) + ROW(mu, k, S::R_H1 + row)); }
                else if (row < ns + nl) m = ROW(mu, k, S::R_LIN + row - ns);
                else m = -ROW(mu, k, S::R_SOC + row - ns - nl);
                for (int j = 0; j < nz; j++) zk[j] += Kl(k)[row * nz + j] * m;
                for (int j = 0; j < np; j++) pv[j] += Kp(k)[row * npa + j] * m;
            }
        }
        double* pv = out + (long)N * (nz + AS);
        for (int which = 0; which < 2; which++) {
            const int nb = which == 0 ? nic : ntc, r0 = which == 0 ? S::G_IC0 : S::G_TC0, r1 = which == 0 ? S::G_IC1 : S::G_TC1;
            const double* H = G() + (which == 0 ? S::Q_H0 : S::Q_HF);
            const double* K = G() + (which == 0 ? S::Q_K0 : S::Q_KF);
            double* zk = out + (long)(which == 0 ? 0 : N - 1) * nz;
            for (int i = 0; i < nb; i++) {
                const double d = GROW(mu, r0 + i) - GROW(mu, r1 + i);
                for (int j = 0; j < nx; j++) zk[j] += H[i * nx + j] * d;
                for (int j = 0; j < np; j++) pv[j] += K[i * npa + j] * d;
                GAUX(out, (which == 0 ? S::GA_YIC : S::GA_YTC) + i) = -(GROW(mu, r0 + i) + GROW(mu, r1 + i));
            }
        }
        for (int j = 0; j < np; j++) {
            pv[j] += GROW(mu, S::G_TRP0 + j) - GROW(mu, S::G_TRP1 + j);
            GAUX(out, S::GA_EP) -= GROW(mu, S::G_TRP0 + j) + GROW(mu, S::G_TRP1 + j);
            for (int i = 0; i < ng; i++) pv[j] += G()[S::Q_LP + i * npa + j] * GROW(mu, S::G_LIN + i);
        }
    }
    void build_constants(double* hn, double* cv, double* qd) const
    {
        std::fill(cv, cv + XI, 0.0); std::fill(qd, qd + XI, 0.0); std::fill(hn, hn + ROWS, 0.0);
        const double ttrp = P[o.scal + 0];
        for (int k = 0; k < N; k++) {
            const double* Pk = st(k);
            for (int r = 0; r < RS; r++) {
                double c = 0.0;
                if (r < 2 * nx) { if (k < N - 1) { const double v = Pk[S::O_CD + r % nx]; c = r < nx ? v : -v; } }
                else if (r < S::R_H1) c = Pk[S::O_CL + (r - S::R_H0)];
                else if (r < S::R_TR0) c = 0.0;
                else if (r < S::R_LIN) { const int j = (r - S::R_TR0) % nz; const double v = Pk[S::O_ZREF + j]; c = r < S::R_TR1 ? -v : v; }
                else if (r < S::R_SOC) c = Pk[S::O_CL + ns + (r - S::R_LIN)];
                else c = -Pk[S::O_CL + ns + nl + (r - S::R_SOC)];
                ROW(hn, k, r) = c;
            }
            for (int j = 0; j < nz; j++) { Z(cv, k, j) = Pk[S::O_Q + j]; Z(qd, k, j) = Pk[S::O_QD + j]; }
            for (int i = 0; i < AS; i++) {
                double c;
                if (i < nx) c = k < N - 1 ? Pk[S::O_OM + i] : 0.0;
                else if (i < nx + ns) c = Pk[S::O_HW + i - nx];
                else c = Pk[S::O_TTR];
                AUX(cv, k, i) = c;
            }
        }
        for (int r = 0; r < RG; r++) {
            double c;
            if (r < S::G_TC0) { const double v = G()[S::Q_L0 + r % (nic > 0 ? nic : 1)]; c = r < S::G_IC1 ? v : -v; }
            else if (r < S::G_TRP0) { const double v = G()[S::Q_LF + (r - S::G_TC0) % (ntc > 0 ? ntc : 1)]; c = r < S::G_TC1 ? v : -v; }
            else if (r < S::G_LIN) { const double v = G()[S::Q_PREF + (r - S::G_TRP0) % (np > 0 ? np : 1)]; c = r < S::G_TRP1 ? -v : v; }
            else c = G()[S::Q_LPC + r - S::G_LIN];
            GROW(hn, r) = c;
        }
        for (int j = 0; j < np; j++) { PV(cv, j) = G()[S::Q_QPL + j]; PV(qd, j) = G()[S::Q_QP + j]; }
        for (int i = 0; i < AG; i++) GAUX(cv, i) = i < nic ? G()[S::Q_BW0 + i] : (i < nic + ntc ? G()[S::Q_BWF + i - nic] : (np > 0 ? ttrp : 0.0));
    }

    // nu-row c of node k: order [dyn nx | hinge ns | bc nbc]; returns false for absent rows
    bool nu_live(int k, int c) const
    {
        if (c < nx) return k < N - 1;
        if (c < nx + ns) return true;
        const int i = c - nx - ns;
        return (k == 0 && i < nic) || (k == N - 1 && i < ntc);
    }
    void nu_row(int k, int c, double* w, double* rt, double* rxv, double& w1, double& w2, double& t1, double& t2, double& rxa,
                bool& hinge) const
    {
        hinge = false;
        if (c < nx) { w1 = ROW(w, k, c); w2 = ROW(w, k, nx + c); t1 = ROW(rt, k, c); t2 = ROW(rt, k, nx + c); rxa = AUX(rxv, k, S::A_Y + c); }
        else if (c < nx + ns) { const int i = c - nx; hinge = true; w1 = ROW(w, k, S::R_H0 + i); w2 = ROW(w, k, S::R_H1 + i); t1 = ROW(rt, k, S::R_H0 + i); t2 = ROW(rt, k, S::R_H1 + i); rxa = AUX(rxv, k, S::A_V + i); }
        else {
            const int i = c - nx - ns;
            const int r0 = k == 0 ? S::G_IC0 : S::G_TC0, r1 = k == 0 ? S::G_IC1 : S::G_TC1, ga = k == 0 ? S::GA_YIC : S::GA_YTC;
            w1 = GROW(w, r0 + i); w2 = GROW(w, r1 + i); t1 = GROW(rt, r0 + i); t2 = GROW(rt, r1 + i); rxa = GAUX(rxv, ga + i);
        }
    }
    static void typeB(const double* w1, const double* w2, int n, int stride, double* T, int ldT)
    {
        double Wt = 0.0;
        for (int j = 0; j < n; j++) Wt += w1[j * stride] + w2[j * stride];
        for (int a = 0; a < n; a++)
            for (int b = 0; b < n; b++) {
                const double ha = w1[a * stride] - w2[a * stride];
                double v;
                if (a != b) v = -ha * (w1[b * stride] - w2[b * stride]) / Wt;
                else {
                    const double d = w1[a * stride] + w2[a * stride];
                    v = 4.0 * w1[a * stride] * w2[a * stride] / d + ha * ha * (Wt - d) / (d * Wt);
                }
                T[a * ldT + b] += v;
            }
    }

    // ---------------- factor ----------------
    bool factor(double* w)
    {
        double Dp[npa * npa];
        for (int i = 0; i < npa * npa; i++) Dp[i] = 0.0;
        std::vector<double> ct((size_t)MNU * npa, 0.0), ctn((size_t)MNU * npa, 0.0);
        for (int k = 0; k < N; k++) {
            double* Sz = &Lz[(size_t)k * nz * nz];
            double* dt = &Dt[(size_t)k * MNU * nz];
            double* ftk = &Ft[(size_t)k * MNU * npa];
            double* cfk = &cf[(size_t)k * MNU * 2];
            std::fill(Sz, Sz + nz * nz, 0.0); std::fill(dt, dt + MNU * nz, 0.0); std::fill(ftk, ftk + MNU * npa, 0.0);
            for (int j = 0; j < nz; j++) Sz[j * nz + j] = st(k)[S::O_QD + j];
            typeB(&ROW(w, k, S::R_TR0), &ROW(w, k, S::R_TR1), nx, 1, Sz, nz);
            typeB(&ROW(w, k, S::R_TR0 + nx), &ROW(w, k, S::R_TR1 + nx), nu, 1, Sz + nx * nz + nx, nz);
            double* c0 = &C0[(size_t)k * nz * npa];
            std::fill(c0, c0 + nz * npa, 0.0);
            for (int i = 0; i < nl; i++) {
                const double wi = ROW(w, k, S::R_LIN + i);
                const double* kr = Kl(k) + (ns + i) * nz; const double* kp = Kp(k) + (ns + i) * npa;
                for (int a = 0; a < nz; a++) {
                    if (kr[a] == 0.0) continue;
                    for (int b = 0; b < nz; b++) Sz[a * nz + b] += wi * kr[a] * kr[b];
                    for (int j = 0; j < np; j++) c0[a * npa + j] += wi * kr[a] * kp[j];
                }
                for (int p1 = 0; p1 < np; p1++) for (int p2 = 0; p2 < np; p2++) Dp[p1 * npa + p2] += wi * kp[p1] * kp[p2];
            }
            for (int c = 0; c < nsoc; c++) {
                const double* Wi = &socW[((size_t)k * NSOC1 + c) * 36 + 16];
                double Ys[4 * nz];
                for (int r = 0; r < 4; r++)
                    for (int j = 0; j < nz; j++) {
                        double acc = 0.0;
                        for (int q = 0; q < 4; q++) acc += Wi[r * 4 + q] * Kl(k)[(ns + nl + 4 * c + q) * nz + j];
                        Ys[r * nz + j] = acc;
                    }
                for (int a = 0; a < nz; a++) for (int b = 0; b < nz; b++) { double acc = 0.0; for (int r = 0; r < 4; r++) acc += Ys[r * nz + a] * Ys[r * nz + b]; Sz[a * nz + b] += acc; }
            }
            if (k > 0) {
                const double* Xp = &X[(size_t)(k - 1) * MNU * nz];
                for (int a = 0; a < nz; a++) for (int b = 0; b < nz; b++) { double acc = 0.0; for (int r = 0; r < MNU; r++) acc += Xp[r * nz + a] * Xp[r * nz + b]; Sz[a * nz + b] += acc; }
            }
            // arrow column right-hand side Cz = C0 + X_{k-1}' ct_{k-1}
            double Cz[nz * npa];
            for (int a = 0; a < nz; a++) for (int j = 0; j < npa; j++) {
                double acc = c0[a * npa + j];
                if (k > 0 && j < np) { const double* Xp = &X[(size_t)(k - 1) * MNU * nz]; for (int r = 0; r < MNU; r++) acc += Xp[r * nz + a] * ct[r * npa + j]; }
                Cz[a * npa + j] = acc;
            }
            if (!chol(Sz, nz, nz)) return false;
            // nu rows: Dt, Ft, coefficients
            for (int c = 0; c < MNU; c++) {
                const bool lv = nu_live(k, c);
                double w1 = 1, w2 = 1, t1, t2, rxa; bool hg = false;
                if (lv) {
                    nu_row(k, c, w, w, w, w1, w2, t1, t2, rxa, hg);
                    if (c < nx) { for (int j = 0; j < nz; j++) dt[c * nz + j] = Dm(k)[c * nz + j]; for (int j = 0; j < np; j++) ftk[c * npa + j] = Fp(k)[c * npa + j]; }
                    else if (c < nx + ns) { for (int j = 0; j < nz; j++) dt[c * nz + j] = Kl(k)[(c - nx) * nz + j]; for (int j = 0; j < np; j++) ftk[c * npa + j] = Kp(k)[(c - nx) * npa + j]; }
                    else {
                        const int i = c - nx - ns;
                        const double* H = G() + (k == 0 ? S::Q_H0 : S::Q_HF); const double* K = G() + (k == 0 ? S::Q_K0 : S::Q_KF);
                        for (int j = 0; j < nx; j++) dt[c * nz + j] = H[i * nx + j];
                        for (int j = 0; j < np; j++) ftk[c * npa + j] = K[i * npa + j];
                    }
                }
                const double iWt = 1.0 / (w1 + w2), kap = (hg ? 1.0 : 4.0) * w1 * w2 * iWt;
                cfk[c * 2 + 0] = lv ? (hg ? w1 : (w1 - w2)) * iWt : 0.0;
                cfk[c * 2 + 1] = lv ? 1.0 / kap : 1.0;
            }
            // Y = Lz^-1 Dt'  (nz x MNU), cb = Lz^-1 Cz
            double* Yk = &Y[(size_t)k * nz * MNU];
            for (int c = 0; c < MNU; c++) {
                double col[nz];
                for (int j = 0; j < nz; j++) col[j] = dt[c * nz + j];
                lsolve(Sz, nz, nz, col);
                for (int j = 0; j < nz; j++) Yk[j * MNU + c] = col[j];
            }
            double cb[nz * npa];
            for (int j = 0; j < np; j++) {
                double col[nz];
                for (int a = 0; a < nz; a++) col[a] = Cz[a * npa + j];
                lsolve(Sz, nz, nz, col);
                for (int a = 0; a < nz; a++) { cb[a * npa + j] = col[a]; Ycz[(size_t)k * nz * npa + a * npa + j] = col[a]; }
            }
            double* Sn = &Lnu[(size_t)k * MNU * MNU];
            for (int c1 = 0; c1 < MNU; c1++)
                for (int c2 = 0; c2 < MNU; c2++) {
                    double acc = 0.0;
                    for (int j = 0; j < nz; j++) acc += Yk[j * MNU + c1] * Yk[j * MNU + c2];
                    if (c1 == c2) acc += cfk[c1 * 2 + 1] + (nu_live(k, c1) ? opt.reg : 0.0);
                    Sn[c1 * MNU + c2] = acc;
                }
            if (!chol(Sn, MNU, MNU)) return false;
            // X = Lnu^-1 Et (Et: dynamics rows only), ct = Lnu^-1 (Ft - Y' cb)
            double* Xk = &X[(size_t)k * MNU * nz];
            for (int j = 0; j < nz; j++) {
                double col[MNU];
                for (int c = 0; c < MNU; c++) col[c] = (c < nx && k < N - 1) ? Em(k)[c * nz + j] : 0.0;
                lsolve(Sn, MNU, MNU, col);
                for (int c = 0; c < MNU; c++) Xk[c * nz + j] = col[c];
            }
            for (int j = 0; j < np; j++) {
                double col[MNU];
                for (int c = 0; c < MNU; c++) { double v = ftk[c * npa + j]; for (int a = 0; a < nz; a++) v -= Yk[a * MNU + c] * cb[a * npa + j]; col[c] = v; }
                lsolve(Sn, MNU, MNU, col);
                for (int c = 0; c < MNU; c++) { ctn[c * npa + j] = col[c]; Ycnu[(size_t)k * MNU * npa + c * npa + j] = col[c]; }
            }
            ct.swap(ctn);
        }
        if (np > 0) {
            // backward sweep of the arrow columns (in place in Ycz / Ycnu)
            for (int j = 0; j < np; j++) {
                double zn[nz];
                for (int a = 0; a < nz; a++) zn[a] = 0.0;
                for (int k = N - 1; k >= 0; k--) {
                    double u[MNU], v[nz];
                    const double* Xk = &X[(size_t)k * MNU * nz]; const double* Yk = &Y[(size_t)k * nz * MNU];
                    for (int c = 0; c < MNU; c++) { double acc = -Ycnu[(size_t)k * MNU * npa + c * npa + j]; if (k < N - 1) for (int a = 0; a < nz; a++) acc += Xk[c * nz + a] * zn[a]; u[c] = acc; }
                    ltsolve(&Lnu[(size_t)k * MNU * MNU], MNU, MNU, u);
                    for (int a = 0; a < nz; a++) { double acc = Ycz[(size_t)k * nz * npa + a * npa + j]; for (int c = 0; c < MNU; c++) acc -= Yk[a * MNU + c] * u[c]; v[a] = acc; }
                    ltsolve(&Lz[(size_t)k * nz * nz], nz, nz, v);
                    for (int c = 0; c < MNU; c++) Ycnu[(size_t)k * MNU * npa + c * npa + j] = u[c];
                    for (int a = 0; a < nz; a++) { Ycz[(size_t)k * nz * npa + a * npa + j] = v[a]; zn[a] = v[a]; }
                }
            }
            double Sp[npa * npa];
            for (int i = 0; i < np; i++) for (int j = 0; j < np; j++) {
                double v = Dp[i * npa + j] + (i == j ? G()[S::Q_QP + i] : 0.0);
                for (int q = 0; q < ng; q++) v += GROW(w, S::G_LIN + q) * G()[S::Q_LP + q * npa + i] * G()[S::Q_LP + q * npa + j];
                for (int k = 0; k < N; k++) {
                    for (int a = 0; a < nz; a++) v -= C0[(size_t)k * nz * npa + a * npa + i] * Ycz[(size_t)k * nz * npa + a * npa + j];
                    for (int c = 0; c < MNU; c++) v -= Ft[(size_t)k * MNU * npa + c * npa + i] * Ycnu[(size_t)k * MNU * npa + c * npa + j];
                }
                Sp[i * npa + j] = v;
            }
            typeB(&GROW(w, S::G_TRP0), &GROW(w, S::G_TRP1), np, 1, Sp, npa);
            if (!chol(Sp, np, npa)) return false;
            for (int i = 0; i < npa * npa; i++) spL[i] = Sp[i];
        }
        return true;
    }

    // ---------------- newton solve: main part of dxi and nu ----------------
    void newton(double* w, double* rtil, double* rxv, double* dxi, double* nuv)
    {
        double bp[npa];
        for (int j = 0; j < npa; j++) bp[j] = 0.0;
        double znx[nz];
        for (int j = 0; j < nz; j++) znx[j] = 0.0;
        for (int k = 0; k < N; k++) {
            double tl[4 * NSOC1];
            for (int c = 0; c < nsoc; c++) {
                const double* Wi = &socW[((size_t)k * NSOC1 + c) * 36 + 16];
                double t1[4];
                for (int r = 0; r < 4; r++) { double acc = 0.0; for (int q = 0; q < 4; q++) acc += Wi[r * 4 + q] * ROW(rtil, k, S::R_SOC + 4 * c + q); t1[r] = acc; }
                for (int r = 0; r < 4; r++) { double acc = 0.0; for (int q = 0; q < 4; q++) acc += Wi[r * 4 + q] * t1[q]; tl[4 * c + r] = acc; }
            }
            double b[nz], t[MNU];
            for (int blk = 0; blk < 2; blk++) {
                const int j0 = blk == 0 ? 0 : nx, n = blk == 0 ? nx : nu;
                double Wt = 0.0, rth = -AUX(rxv, k, blk == 0 ? S::A_EX : S::A_EU);
                for (int q = j0; q < j0 + n; q++) { const double w1 = ROW(w, k, S::R_TR0 + q), w2 = ROW(w, k, S::R_TR1 + q); Wt += w1 + w2; rth += w1 * ROW(rtil, k, S::R_TR0 + q) + w2 * ROW(rtil, k, S::R_TR1 + q); }
                for (int j = j0; j < j0 + n; j++) {
                    const double w1 = ROW(w, k, S::R_TR0 + j), w2 = ROW(w, k, S::R_TR1 + j);
                    double acc = -Z(rxv, k, j) + znx[j];
                    acc += -(w1 * ROW(rtil, k, S::R_TR0 + j) - w2 * ROW(rtil, k, S::R_TR1 + j)) + (w1 - w2) * rth / Wt;
                    for (int i = 0; i < nl; i++) acc += Kl(k)[(ns + i) * nz + j] * (-ROW(w, k, S::R_LIN + i) * ROW(rtil, k, S::R_LIN + i));
                    for (int r = 0; r < 4 * nsoc; r++) acc += Kl(k)[(ns + nl + r) * nz + j] * tl[r];
                    b[j] = acc;
                }
            }
            const double* cfk = &cf[(size_t)k * MNU * 2];
            for (int c = 0; c < MNU; c++) {
                t[c] = 0.0;
                if (!nu_live(k, c)) continue;
                double w1, w2, t1, t2, rxa; bool hg;
                nu_row(k, c, w, rtil, rxv, w1, w2, t1, t2, rxa, hg);
                const double r1 = w1 * t1, r2 = w2 * t2, rth = -rxa + r1 + r2;
                t[c] = (-(r1 - (hg ? 0.0 : r2)) + cfk[c * 2] * rth) * cfk[c * 2 + 1];
            }
            for (int r = 0; r < nl + 4 * nsoc; r++) {
                const double v = r < nl ? -ROW(w, k, S::R_LIN + r) * ROW(rtil, k, S::R_LIN + r) : tl[r - nl];
                for (int j = 0; j < np; j++) bp[j] += Kp(k)[(ns + r) * npa + j] * v;
            }
            // chain
            lsolve(&Lz[(size_t)k * nz * nz], nz, nz, b);
            const double* Yk = &Y[(size_t)k * nz * MNU]; const double* Xk = &X[(size_t)k * MNU * nz];
            for (int c = 0; c < MNU; c++) { double acc = t[c]; for (int j = 0; j < nz; j++) acc -= Yk[j * MNU + c] * b[j]; t[c] = acc; }
            lsolve(&Lnu[(size_t)k * MNU * MNU], MNU, MNU, t);
            for (int j = 0; j < nz; j++) { double acc = 0.0; for (int c = 0; c < MNU; c++) acc += Xk[c * nz + j] * t[c]; znx[j] = acc; }
            for (int j = 0; j < nz; j++) fb[(size_t)k * nz + j] = b[j];
            for (int c = 0; c < MNU; c++) ft[(size_t)k * MNU + c] = t[c];
        }
        double zn[nz];
        for (int j = 0; j < nz; j++) zn[j] = 0.0;
        for (int k = N - 1; k >= 0; k--) {
            const double* Yk = &Y[(size_t)k * nz * MNU]; const double* Xk = &X[(size_t)k * MNU * nz];
            double u[MNU], v[nz];
            for (int c = 0; c < MNU; c++) { double acc = -ft[(size_t)k * MNU + c]; if (k < N - 1) for (int j = 0; j < nz; j++) acc += Xk[c * nz + j] * zn[j]; u[c] = acc; }
            ltsolve(&Lnu[(size_t)k * MNU * MNU], MNU, MNU, u);
            for (int j = 0; j < nz; j++) { double acc = fb[(size_t)k * nz + j]; for (int c = 0; c < MNU; c++) acc -= Yk[j * MNU + c] * u[c]; v[j] = acc; }
            ltsolve(&Lz[(size_t)k * nz * nz], nz, nz, v);
            for (int j = 0; j < nz; j++) { Z(dxi, k, j) = v[j]; zn[j] = v[j]; }
            for (int c = 0; c < MNU; c++) nuv[(size_t)k * MNU + c] = u[c];
        }
        double dp[npa];
        for (int j = 0; j < npa; j++) dp[j] = 0.0;
        if (np > 0) {
            for (int k = 0; k < N; k++) {
                for (int a = 0; a < nz; a++) for (int j = 0; j < np; j++) bp[j] -= C0[(size_t)k * nz * npa + a * npa + j] * Z(dxi, k, a);
                for (int c = 0; c < MNU; c++) for (int j = 0; j < np; j++) bp[j] -= Ft[(size_t)k * MNU * npa + c * npa + j] * nuv[(size_t)k * MNU + c];
            }
            double Wt = 0.0, rth = -GAUX(rxv, S::GA_EP);
            for (int q = 0; q < np; q++) { const double w1 = GROW(w, S::G_TRP0 + q), w2 = GROW(w, S::G_TRP1 + q); Wt += w1 + w2; rth += w1 * GROW(rtil, S::G_TRP0 + q) + w2 * GROW(rtil, S::G_TRP1 + q); }
            for (int j = 0; j < np; j++) {
                const double w1 = GROW(w, S::G_TRP0 + j), w2 = GROW(w, S::G_TRP1 + j);
                double v = bp[j] - PV(rxv, j);
                v += -(w1 * GROW(rtil, S::G_TRP0 + j) - w2 * GROW(rtil, S::G_TRP1 + j)) + (w1 - w2) * rth / Wt;
                for (int q = 0; q < ng; q++) v += G()[S::Q_LP + q * npa + j] * (-GROW(w, S::G_LIN + q) * GROW(rtil, S::G_LIN + q));
                dp[j] = v;
            }
            lsolve(spL.data(), np, npa, dp); ltsolve(spL.data(), np, npa, dp);
            for (int k = 0; k < N; k++) {
                for (int a = 0; a < nz; a++) for (int j = 0; j < np; j++) Z(dxi, k, a) -= Ycz[(size_t)k * nz * npa + a * npa + j] * dp[j];
                for (int c = 0; c < MNU; c++) for (int j = 0; j < np; j++) nuv[(size_t)k * MNU + c] -= Ycnu[(size_t)k * MNU * npa + c * npa + j] * dp[j];
            }
        }
        for (int j = 0; j < npa; j++) PV(dxi, j) = dp[j];
    }

    // ---------------- finish: aux steps, gd = G dxi, dl ----------------
    void finish(double* w, double* rtil, double* rxv, double* dxi, double* nuv, double* gd, double* dl)
    {
        for (int k = 0; k < N; k++) {
            double arow[RS];
            for (int r = 0; r < RS; r++) arow[r] = row_main(dxi, k, r);
            // type A / hinge aux
            for (int c = 0; c < nx + ns; c++) {
                double val = 0.0;
                if (nu_live(k, c)) {
                    double w1, w2, t1, t2, rxa; bool hg;
                    nu_row(k, c, w, rtil, rxv, w1, w2, t1, t2, rxa, hg);
                    const double rth = -rxa + w1 * t1 + w2 * t2;
                    const double av = c < nx ? arow[c] : arow[S::R_H0 + c - nx];
                    val = (rth + (hg ? w1 : (w1 - w2)) * av) / (w1 + w2);
                }
                AUX(dxi, k, c) = val;
            }
            for (int blk = 0; blk < 2; blk++) {
                const int j0 = blk == 0 ? 0 : nx, n = blk == 0 ? nx : nu;
                double Wt = 0.0, rth = -AUX(rxv, k, blk == 0 ? S::A_EX : S::A_EU), ha = 0.0;
                for (int q = j0; q < j0 + n; q++) {
                    const double w1 = ROW(w, k, S::R_TR0 + q), w2 = ROW(w, k, S::R_TR1 + q);
                    Wt += w1 + w2; rth += w1 * ROW(rtil, k, S::R_TR0 + q) + w2 * ROW(rtil, k, S::R_TR1 + q); ha += (w1 - w2) * Z(dxi, k, q);
                }
                AUX(dxi, k, blk == 0 ? S::A_EX : S::A_EU) = (rth + ha) / Wt;
            }
            for (int r = 0; r < RS; r++) {
                double g, d;
                if (r < 2 * nx) {
                    const int i = r % nx;
                    if (k < N - 1) { g = arow[r] - AUX(dxi, k, S::A_Y + i); const double nv = nuv[(size_t)k * MNU + i], rxa = AUX(rxv, k, S::A_Y + i); d = r < nx ? 0.5 * (rxa + nv) : 0.5 * (rxa - nv); }
                    else { g = 0.0; d = 0.0; }
                } else if (r < S::R_TR0) {
                    const int i = (r - S::R_H0) % NS1;
                    g = arow[r] - AUX(dxi, k, S::A_V + i);
                    const double nv = nuv[(size_t)k * MNU + nx + i], rxa = AUX(rxv, k, S::A_V + i);
                    d = r < S::R_H1 ? nv : rxa - nv;
                } else if (r < S::R_LIN) {
                    const int j = (r - S::R_TR0) % nz;
                    g = arow[r] - AUX(dxi, k, j < nx ? S::A_EX : S::A_EU);
                    d = ROW(w, k, r) * (g + ROW(rtil, k, r));
                } else if (r < S::R_SOC) { g = arow[r]; d = ROW(w, k, r) * (g + ROW(rtil, k, r)); }
                else {
                    g = arow[r];
                    const int c = (r - S::R_SOC) / 4, rr = (r - S::R_SOC) % 4;
                    const double* Wi = &socW[((size_t)k * NSOC1 + c) * 36 + 16];
                    double t1[4];
                    for (int q = 0; q < 4; q++) { double acc = 0.0; for (int q2 = 0; q2 < 4; q2++) acc += Wi[q * 4 + q2] * (arow[S::R_SOC + 4 * c + q2] + ROW(rtil, k, S::R_SOC + 4 * c + q2)); t1[q] = acc; }
                    double acc = 0.0;
                    for (int q = 0; q < 4; q++) acc += Wi[rr * 4 + q] * t1[q];
                    d = acc;
                }
                ROW(gd, k, r) = g; ROW(dl, k, r) = d;
            }
        }
        for (int which = 0; which < 2; which++) {
            const int nb = which == 0 ? nic : ntc, k = which == 0 ? 0 : N - 1;
            if (which == 1 && N == 1) { /* single node: both sets live at node 0 -- not used (N >= 2) */ }
            for (int i = 0; i < nb; i++) {
                const int c = nx + ns + i;
                double w1, w2, t1, t2, rxa; bool hg;
                nu_row(k, c, w, rtil, rxv, w1, w2, t1, t2, rxa, hg);
                const double rth = -rxa + w1 * t1 + w2 * t2, av = bc_main(dxi, which, i);
                const double dy = (rth + (w1 - w2) * av) / (w1 + w2), nv = nuv[(size_t)k * MNU + c];
                GAUX(dxi, (which == 0 ? S::GA_YIC : S::GA_YTC) + i) = dy;
                const int r0 = which == 0 ? S::G_IC0 : S::G_TC0, r1 = which == 0 ? S::G_IC1 : S::G_TC1;
                GROW(gd, r0 + i) = av - dy; GROW(gd, r1 + i) = -av - dy;
                GROW(dl, r0 + i) = 0.5 * (rxa + nv); GROW(dl, r1 + i) = 0.5 * (rxa - nv);
            }
        }
        double detap = 0.0;
        if (np > 0) {
            double Wt = 0.0, rth = -GAUX(rxv, S::GA_EP), ha = 0.0;
            for (int q = 0; q < np; q++) { const double w1 = GROW(w, S::G_TRP0 + q), w2 = GROW(w, S::G_TRP1 + q); Wt += w1 + w2; rth += w1 * GROW(rtil, S::G_TRP0 + q) + w2 * GROW(rtil, S::G_TRP1 + q); ha += (w1 - w2) * PV(dxi, q); }
            detap = (rth + ha) / Wt;
        }
        GAUX(dxi, S::GA_EP) = detap;
        for (int j = 0; j < np; j++) {
            const double g0 = PV(dxi, j) - detap, g1 = -PV(dxi, j) - detap;
            GROW(gd, S::G_TRP0 + j) = g0; GROW(gd, S::G_TRP1 + j) = g1;
            GROW(dl, S::G_TRP0 + j) = GROW(w, S::G_TRP0 + j) * (g0 + GROW(rtil, S::G_TRP0 + j));
            GROW(dl, S::G_TRP1 + j) = GROW(w, S::G_TRP1 + j) * (g1 + GROW(rtil, S::G_TRP1 + j));
        }
        for (int i = 0; i < ng; i++) {
            double acc = 0.0;
            for (int j = 0; j < np; j++) acc += G()[S::Q_LP + i * npa + j] * PV(dxi, j);
            GROW(gd, S::G_LIN + i) = acc;
            GROW(dl, S::G_LIN + i) = GROW(w, S::G_LIN + i) * (acc + GROW(rtil, S::G_LIN + i));
        }
    }

    bool nt_update(double* s, double* lam)
    {
        bool ok = true;
        for (int k = 0; k < N; k++)
            for (int c = 0; c < nsoc; c++) {
                double sv[4], zv[4];
                for (int q = 0; q < 4; q++) { sv[q] = ROW(s, k, S::R_SOC + 4 * c + q); zv[q] = ROW(lam, k, S::R_SOC + 4 * c + q); }
                const double sres = std::sqrt(sv[0] * sv[0] - sv[1] * sv[1] - sv[2] * sv[2] - sv[3] * sv[3]);
                const double zres = std::sqrt(zv[0] * zv[0] - zv[1] * zv[1] - zv[2] * zv[2] - zv[3] * zv[3]);
                double sb[4], zb[4], dot = 0.0;
                for (int q = 0; q < 4; q++) { sb[q] = sv[q] / sres; zb[q] = zv[q] / zres; dot += sb[q] * zb[q]; }
                const double gamma = std::sqrt((1.0 + dot) / 2.0);
                double wb[4];
                wb[0] = (sb[0] + zb[0]) / (2 * gamma);
                for (int q = 1; q < 4; q++) wb[q] = (sb[q] - zb[q]) / (2 * gamma);
                const double eta = std::sqrt(sres / zres);
                double* Wm = &socW[((size_t)k * NSOC1 + c) * 36];
                for (int r = 0; r < 4; r++)
                    for (int q = 0; q < 4; q++) {
                        double v;
                        if (r == 0) v = wb[q]; else if (q == 0) v = wb[r]; else v = (r == q ? 1.0 : 0.0) + wb[r] * wb[q] / (1.0 + wb[0]);
                        Wm[r * 4 + q] = v * eta;
                        Wm[16 + r * 4 + q] = ((r == 0) != (q == 0) ? -v : v) / eta;
                    }
                for (int r = 0; r < 4; r++) { double acc = 0.0; for (int q = 0; q < 4; q++) acc += Wm[r * 4 + q] * zv[q]; Wm[32 + r] = acc; }
                if (!(sres > 0.0) || !(zres > 0.0) || !std::isfinite(eta)) ok = false;
            }
        return ok;
    }
    static double soc_step(const double* s, const double* d)
    {
        const double s0 = s[0], d0 = d[0];
        const double dd = d[1] * d[1] + d[2] * d[2] + d[3] * d[3], sd = s[1] * d[1] + s[2] * d[2] + s[3] * d[3], ss = s[1] * s[1] + s[2] * s[2] + s[3] * s[3];
        const double qa = d0 * d0 - dd, qb = 2.0 * (s0 * d0 - sd), qc = s0 * s0 - ss;
        double r1 = -1.0, r2 = -1.0;
        if (std::fabs(qa) <= 1e-14 * (d0 * d0 + dd + 1e-300)) { if (qb < 0.0) r1 = -qc / qb; }
        else {
            const double disc = qb * qb - 4.0 * qa * qc;
            if (disc >= 0.0) { const double sq = std::sqrt(disc), qq = -0.5 * (qb + (qb >= 0.0 ? sq : -sq)); r1 = qq / qa; if (qq != 0.0) r2 = qc / qq; }
        }
        double am = 1e300;
        if (r1 > 0.0 && s0 + r1 * d0 >= -1e-12 * (std::fabs(s0) + std::fabs(r1 * d0))) am = std::min(am, r1);
        if (r2 > 0.0 && s0 + r2 * d0 >= -1e-12 * (std::fabs(s0) + std::fabs(r2 * d0))) am = std::min(am, r2);
        return am;
    }
    double min_margin(double* v) const
    {
        double mm = 1e300;
        for (long i = 0; i < ROWS; i++) {
            if (is_dead(i)) continue;
            if (!is_soc(i)) mm = std::min(mm, v[i]);
            else if (((i % RS) - S::R_SOC) % 4 == 0) mm = std::min(mm, v[i] - std::sqrt(v[i + 1] * v[i + 1] + v[i + 2] * v[i + 2] + v[i + 3] * v[i + 3]));
        }
        return mm;
    }

    // ---------------- main loop (oracle/ipm_struct.py::solve, csrc/ipm2_run.hpp) ----------------
    IpmResult solve(std::vector<double>& best)
    {
        std::vector<double> xi(XI, 0), dxi(XI, 0), rx(XI, 0), exi(XI, 0), rxe(XI, 0), cv(XI), qd(XI);
        std::vector<double> s(ROWS, 0), lam(ROWS, 0), rz(ROWS, 0), w(ROWS, 1), rtil(ROWS, 0), ds(ROWS, 0), dl(ROWS, 0), gd(ROWS, 0), r2(ROWS, 0),
            el(ROWS, 0), hneg(ROWS), ge(ROWS, 0), sn(ROWS), ln(ROWS);
        std::vector<double> nuv((size_t)N * MNU, 0);
        best.assign(XI, 0.0);
        build_constants(hneg.data(), cv.data(), qd.data());
        const double cost_const = P[o.scal + 1];
        double nh = 0, nc = 0, deg = 0;
        for (long i = 0; i < ROWS; i++) { nh += hneg[i] * hneg[i]; if (!is_dead(i) && !is_soc(i)) deg += 1.0; }
        for (long i = 0; i < XI; i++) nc += cv[i] * cv[i];
        deg += (double)N * nsoc;
        const double nrm_h = std::max(1.0, std::sqrt(nh)), nrm_c = std::max(1.0, std::sqrt(nc));
        IpmResult res, bestr;
        double best_merit = 1e300; int best_it = 0;
        double prog_merit = 1e300; int prog_it = 0;
        double gap = 0, mu = 0, sigma = 0, relgap_it = 1e300;
        int it;
        int it0 = -1;
        bool snap_taken = false;
        double trace_alpha = 0.0;
        bool snapA_taken = false;
        bool sn_taken[IpmOpts::NL] = {false, false, false, false};
        double mu_prev_it = 1e300;     // mu of the previous iterate of this solve (a cold solve comes from above every level)
        if (use_warm && opt.warm == 5 && sn_ok[snap_level] && (long)xi_sn[snap_level].size() == XI) {
            it0 = 0;
            xi = xi_sn[snap_level]; s = s_sn[snap_level]; lam = lam_sn[snap_level];
        } else
        if (use_warm && opt.warm == 4 && snap_level == 0 && snapA_ok && (long)xi_snapA.size() == XI) {
            it0 = 0;
            xi = xi_snapA; s = s_snapA; lam = lam_snapA;
        } else
        if (use_warm && (opt.warm == 3 || (opt.warm == 4 && snap_level == 1)) && snap_ok && (long)xi_snap.size() == XI) {
            // the well-centred intermediate iterate of the previous solve, as it is: the infeasible-start iteration absorbs the
            // change of the problem data (residuals of the order of the reference deviation)
            it0 = 0;
            xi = xi_snap; s = s_snap; lam = lam_snap;
        } else
        if (use_warm && opt.warm == 1 && (long)xi_prev.size() != XI) xi_prev.assign(XI, 0.0);   // experiment: structured COLD start
        if (use_warm && opt.warm != 3 && opt.warm != 4 && opt.warm != 5 && (long)xi_prev.size() == XI) {
            it0 = 0;
            const double m0 = opt.warm_mu;
            // primal: the reference point (= previous solution) with its epigraph variables
            xi = xi_prev;
            for (int k = 0; k < N; k++) for (int j = 0; j < nz; j++) Z(xi.data(), k, j) = st(k)[S::O_ZREF + j];
            for (int j = 0; j < np; j++) PV(xi.data(), j) = G()[S::Q_PREF + j];
            if (opt.warm == 1) {
                // structured point: epigraph variables chosen so that every pair / L_inf block is exactly centred at m0
                std::vector<double> a0(ROWS, 0.0), zero(XI, 0.0);
                std::vector<double> xm = xi;
                for (int k = 0; k < N; k++) for (int i = 0; i < AS; i++) AUX(xm.data(), k, i) = 0.0;
                for (int i = 0; i < AG; i++) GAUX(xm.data(), i) = 0.0;
                G_apply(xm.data(), a0.data());
                for (long i = 0; i < ROWS; i++) a0[i] += hneg[i];   // row activity incl. constants (first row of each pair: +a)
                auto typeA = [&](double a, double om) { om = std::max(om, 1e-300); return (m0 + std::sqrt(m0 * m0 + om * om * a * a)) / om; };
                for (int k = 0; k < N; k++) {
                    for (int i = 0; i < nx; i++) AUX(xi.data(), k, S::A_Y + i) = k < N - 1 ? typeA(ROW(a0.data(), k, i), st(k)[S::O_OM + i]) : 0.0;
                    for (int i = 0; i < ns; i++) {
                        const double hw = st(k)[S::O_HW + i], ah = ROW(a0.data(), k, S::R_H0 + i), bq = hw * ah + 2 * m0;
                        AUX(xi.data(), k, S::A_V + i) = (bq + std::sqrt(bq * bq - 4 * hw * m0 * ah)) / (2 * hw);
                    }
                    AUX(xi.data(), k, S::A_EX) = 2 * nx * m0 / st(k)[S::O_TTR]; AUX(xi.data(), k, S::A_EU) = 2 * nu * m0 / st(k)[S::O_TTR];
                }
                for (int i = 0; i < nic; i++) GAUX(xi.data(), S::GA_YIC + i) = typeA(GROW(a0.data(), S::G_IC0 + i), G()[S::Q_BW0 + i]);
                for (int i = 0; i < ntc; i++) GAUX(xi.data(), S::GA_YTC + i) = typeA(GROW(a0.data(), S::G_TC0 + i), G()[S::Q_BWF + i]);
                GAUX(xi.data(), S::GA_EP) = np > 0 ? 2 * np * m0 / P[o.scal + 0] : 0.0;
            }
            G_apply(xi.data(), gd.data());
            for (long i = 0; i < ROWS; i++) s[i] = -(gd[i] + hneg[i]);
            const double fl = std::sqrt(m0);
            for (long i = 0; i < ROWS; i++) {
                if (is_dead(i)) { s[i] = 1.0; lam[i] = 1.0; continue; }
                if (is_soc(i)) continue;
                if (opt.warm == 1) { s[i] = std::max(s[i], fl * 1e-3 + 0 * fl); s[i] = std::max(s[i], 1e-300); lam[i] = m0 / s[i]; }
                else {
                    double l = std::max(lam_prev[i], 1e-14);
                    // slack pushed up to complementarity m0 with the old multiplier, but never beyond sqrt(m0): a row that was
                    // inactive (multiplier ~ 0) keeps its own slack and gets the multiplier m0 / s instead -- pushing ITS slack to
                    // m0 / l = 1e9 made the warm point wildly primal infeasible (relative residual 1e6) and 3 % of the warm solves
                    // spend 45 iterations without a full step before the cold repeat
                    double sv = std::max(s[i], std::min(m0 / l, fl));
                    l = std::max(l, m0 / sv);
                    s[i] = sv; lam[i] = l;
                }
            }
            for (int k = 0; k < N; k++) for (int c = 0; c < nsoc; c++) {
                const long b0 = (long)k * RS + S::R_SOC + 4 * c;
                double ms = s[b0] - std::sqrt(s[b0 + 1] * s[b0 + 1] + s[b0 + 2] * s[b0 + 2] + s[b0 + 3] * s[b0 + 3]);
                if (ms < fl) s[b0] += fl - ms;
                if (opt.warm == 1) {
                    const double det = s[b0] * s[b0] - s[b0 + 1] * s[b0 + 1] - s[b0 + 2] * s[b0 + 2] - s[b0 + 3] * s[b0 + 3];
                    lam[b0] = m0 * s[b0] / det; for (int q = 1; q < 4; q++) lam[b0 + q] = -m0 * s[b0 + q] / det;
                } else {
                    for (int q = 0; q < 4; q++) lam[b0 + q] = lam_prev[b0 + q];
                    double ml = lam[b0] - std::sqrt(lam[b0 + 1] * lam[b0 + 1] + lam[b0 + 2] * lam[b0 + 2] + lam[b0 + 3] * lam[b0 + 3]);
                    if (ml < fl) lam[b0] += fl - ml;
                }
            }
        }
        for (it = it0; it <= opt.max_iter; it++) {
            if (it < 0) {
                for (long i = 0; i < ROWS; i++) { w[i] = 1.0; rtil[i] = hneg[i]; r2[i] = 0.0; }
                for (long i = 0; i < XI; i++) { rx[i] = cv[i]; xi[i] = 0.0; rxe[i] = 0.0; }
                for (int k = 0; k < N; k++) for (int c = 0; c < nsoc; c++) { double* Wm = &socW[((size_t)k * NSOC1 + c) * 36]; for (int q = 0; q < 16; q++) { Wm[q] = (q % 5 == 0) ? 1.0 : 0.0; Wm[16 + q] = Wm[q]; } for (int q = 0; q < 4; q++) Wm[32 + q] = 0.0; }
            } else {
                GT_apply(lam.data(), rx.data());
                G_apply(xi.data(), gd.data());
                double lrz = 0, nrz = 0, nrx = 0, pc = 0;
                gap = 0;
                for (long i = 0; i < XI; i++) { const double r_ = rx[i] + qd[i] * xi[i] + cv[i]; rx[i] = r_; nrx += r_ * r_; pc += 0.5 * qd[i] * xi[i] * xi[i] + cv[i] * xi[i]; }
                for (long i = 0; i < ROWS; i++) {
                    const bool lv = !is_dead(i);
                    const double val = lv ? gd[i] + s[i] + hneg[i] : 0.0;
                    rz[i] = val; rtil[i] = val - s[i];
                    w[i] = (lv && !is_soc(i)) ? lam[i] / s[i] : 1.0;
                    if (lv) { gap += s[i] * lam[i]; lrz += lam[i] * val; nrz += val * val; }
                }
                const bool snap_skip0 = it0 == 0 && it == 0;   // a warm solve refreshes its snapshots only after a step on the NEW problem
                if (!snap_skip0)
                if (opt.warm >= 3 && !snap_taken && gap / deg <= opt.warm_save_mu) { xi_snap = xi; s_snap = s; lam_snap = lam; snap_taken = true; }
                if (!snap_skip0 && opt.warm == 5)
                    for (int l = 0; l < IpmOpts::NL; l++)
                        if (!sn_taken[l] && !(it0 == 0 && l < snap_level) && gap / deg <= opt.lvl_mu[l] && (!opt.cross || mu_prev_it > opt.lvl_mu[l]) && gap / deg >= opt.lvl_mu[l] * opt.lvl_floor) { xi_sn[l] = xi; s_sn[l] = s; lam_sn[l] = lam; sn_taken[l] = true; }
                mu_prev_it = gap / deg;
                if (!snap_skip0)
                if (opt.warm == 4 && !snapA_taken && gap / deg <= opt.warm_save_mu_coarse) { xi_snapA = xi; s_snapA = s; lam_snapA = lam; snapA_taken = true; }
                const double pcost = pc, dcost = pcost + lrz - gap;
                const double pres = std::sqrt(nrz) / nrm_h, dres = std::sqrt(nrx) / nrm_c;
                const double relgap = pcost < 0.0 ? gap / -pcost : (dcost > 0.0 ? gap / dcost : 1e300);
                relgap_it = relgap;
                const double merit = std::max(std::max(pres / opt.feastol, dres / opt.feastol), std::min(gap / opt.abstol, relgap / opt.reltol));
                res.iters = it;
                if (std::isfinite(merit) && merit < best_merit) {
                    // (progress for the stall rule: an improvement by at least the factor stall_rel -- experiment SCP_CPU_STALL_REL; 1 = any)
                    if (merit < opt.stall_rel * prog_merit) { prog_merit = merit; prog_it = it; }
                    best_merit = merit; best_it = it; best = xi;
                    bestr.pcost = pcost + cost_const; bestr.dcost = dcost + cost_const; bestr.gap = gap; bestr.pres = pres; bestr.dres = dres; bestr.relgap = relgap;
                }
                if (std::getenv("SCP_CPU_TRACE")) std::fprintf(stderr, "  it %2d gap %.3e relgap %.2e pres %.2e dres %.2e mu %.2e sigma %.3f alpha %.4f\n", it, gap, relgap, pres, dres, gap / deg, sigma, trace_alpha);
                if (!std::isfinite(merit)) { res.status = 3; break; }
                if (merit <= 1.0) { res.status = 0; break; }
                if (it == opt.max_iter) break;
                if (it0 == 0 && it >= (opt.warm == 5 ? std::min(opt.warm_max_iter, opt.lvl_cap[snap_level]) : opt.warm_max_iter)) break;   // a warm start that has not converged by now is abandoned
                if (best_merit <= 1e3 && it - (opt.stall_rel < 1.0 ? prog_it : best_it) >= opt.stall) break;
                if (!nt_update(s.data(), lam.data())) { res.status = 3; break; }
                mu = gap / deg;
            }
            {   // experiment (SCP_CPU_REGESC): a factorisation that breaks down is repeated with 10x the static regularisation
                bool fok = factor(w.data());
                for (int tr = 0; !fok && tr < opt.reg_escalate; tr++) { opt.reg *= 10.0; fok = factor(w.data()); }
                if (!fok) { res.status = 3; break; }
            }
            for (int phase = 0; phase < 2; phase++) {
                if (it >= 0 && phase == 1) {
                    for (long i = 0; i < ROWS; i++) {
                        if (is_soc(i)) continue;
                        double val = rz[i] - s[i];
                        if (!is_dead(i)) val += (sigma * mu - ds[i] * dl[i]) / lam[i];
                        rtil[i] = val;
                    }
                    for (int k = 0; k < N; k++) for (int c = 0; c < nsoc; c++) {
                        const long b0 = (long)k * RS + S::R_SOC + 4 * c;
                        const double* Wv = &socW[((size_t)k * NSOC1 + c) * 36]; const double* Wi = Wv + 16; const double* lt = Wv + 32;
                        double u1[4], u2[4], dsv[4], uu[4];
                        for (int q = 0; q < 4; q++) { double a1 = 0, a2 = 0; for (int q2 = 0; q2 < 4; q2++) { a1 += Wi[q * 4 + q2] * ds[b0 + q2]; a2 += Wv[q * 4 + q2] * dl[b0 + q2]; } u1[q] = a1; u2[q] = a2; }
                        dsv[0] = sigma * mu - (lt[0] * lt[0] + lt[1] * lt[1] + lt[2] * lt[2] + lt[3] * lt[3]) - (u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2] + u1[3] * u2[3]);
                        for (int q = 1; q < 4; q++) dsv[q] = -2.0 * lt[0] * lt[q] - (u1[0] * u2[q] + u2[0] * u1[q]);
                        const double den = lt[0] * lt[0] - lt[1] * lt[1] - lt[2] * lt[2] - lt[3] * lt[3];
                        uu[0] = (lt[0] * dsv[0] - lt[1] * dsv[1] - lt[2] * dsv[2] - lt[3] * dsv[3]) / den;
                        for (int q = 1; q < 4; q++) uu[q] = (dsv[q] - uu[0] * lt[q]) / lt[0];
                        for (int q = 0; q < 4; q++) { double acc = 0; for (int q2 = 0; q2 < 4; q2++) acc += Wv[q * 4 + q2] * uu[q2]; rtil[b0 + q] = rz[b0 + q] + acc; }
                    }
                }
                int nref_eff = (it < 0 || !(relgap_it < opt.ref_gap)) ? 0 : opt.nref;
                if (opt.ref_on_stall && it >= 0 && it - best_it >= opt.ref_on_stall) nref_eff = std::max(nref_eff, 1);   // experiment
                if (opt.ref_corrector_only && phase == 0) nref_eff = 0;
                for (int rf = 0; rf <= nref_eff; rf++) {
                    double *rt_ = rtil.data(), *rx_ = rx.data(), *ox = dxi.data(), *og = gd.data(), *ol = dl.data();
                    if (it < 0 && phase == 0) { rx_ = rxe.data(); ox = xi.data(); }
                    if (it < 0 && phase == 1) { rt_ = r2.data(); og = ge.data(); ol = el.data(); }
                    if (rf > 0) {
                        GT_apply(dl.data(), rxe.data());
                        for (long i = 0; i < XI; i++) rxe[i] = rxe[i] + qd[i] * dxi[i] + rx[i];
                        for (long i = 0; i < ROWS; i++) { if (is_soc(i)) continue; r2[i] = is_dead(i) ? 0.0 : rtil[i] + gd[i] - dl[i] / w[i]; }
                        for (int k = 0; k < N; k++) for (int c = 0; c < nsoc; c++) {
                            const long b0 = (long)k * RS + S::R_SOC + 4 * c;
                            const double* Wv = &socW[((size_t)k * NSOC1 + c) * 36];
                            double t1[4];
                            for (int q = 0; q < 4; q++) { double acc = 0; for (int q2 = 0; q2 < 4; q2++) acc += Wv[q * 4 + q2] * dl[b0 + q2]; t1[q] = acc; }
                            for (int rr = 0; rr < 4; rr++) { double acc = 0; for (int q = 0; q < 4; q++) acc += Wv[rr * 4 + q] * t1[q]; r2[b0 + rr] = rtil[b0 + rr] + gd[b0 + rr] - acc; }
                        }
                        rt_ = r2.data(); rx_ = rxe.data(); ox = exi.data(); og = ge.data(); ol = el.data();
                    }
                    newton(w.data(), rt_, rx_, ox, nuv.data());
                    finish(w.data(), rt_, rx_, ox, nuv.data(), og, ol);
                    if (rf > 0) {
                        for (long i = 0; i < XI; i++) dxi[i] += exi[i];
                        for (long i = 0; i < ROWS; i++) { dl[i] += el[i]; gd[i] += ge[i]; }
                    }
                }
                if (it < 0 && phase == 0) { for (long i = 0; i < ROWS; i++) s[i] = -(gd[i] + hneg[i]); }
                else if (it < 0) {
                    for (long i = 0; i < ROWS; i++) lam[i] = ge[i];
                    for (int r = 0; r < 2 * nx; r++) { ROW(lam.data(), N - 1, r) = 1.0; ROW(s.data(), N - 1, r) = 1.0; }
                    static const int init_mode = std::getenv("SCP_CPU_INIT") ? std::atoi(std::getenv("SCP_CPU_INIT")) : 0;
                    auto add_e = [&](double* v, double sh) { for (long i = 0; i < ROWS; i++) { if (is_dead(i)) continue; if (!is_soc(i) || (((i % RS) - S::R_SOC) % 4 == 0)) v[i] += sh; } };
                    if (init_mode == 0) {
                    for (int pass = 0; pass < 2; pass++) {
                        double* v = pass == 0 ? s.data() : lam.data();
                        const double mm = min_margin(v);
                        if (mm <= 0.0) {
                            const double sh = 1.0 - mm;
                            for (long i = 0; i < ROWS; i++) { if (is_dead(i)) continue; if (!is_soc(i) || (((i % RS) - S::R_SOC) % 4 == 0)) v[i] += sh; }
                        }
                    }
                    } else {   // experiment: Mehrotra's starting point (shift by 1.5 x the violation, then by half the complementarity over the other's sum)
                        const double ms = min_margin(s.data()), ml = min_margin(lam.data());
                        add_e(s.data(), std::max(-1.5 * ms, 0.0)); add_e(lam.data(), std::max(-1.5 * ml, 0.0));
                        double sl = 0.0, ss = 0.0, sm = 0.0;
                        for (long i = 0; i < ROWS; i++) { if (is_dead(i)) continue; sl += s[i] * lam[i]; if (!is_soc(i) || (((i % RS) - S::R_SOC) % 4 == 0)) { ss += s[i]; sm += lam[i]; } }
                        add_e(s.data(), 0.5 * sl / std::max(sm, 1e-300)); add_e(lam.data(), 0.5 * sl / std::max(ss, 1e-300));
                        if (init_mode == 2) { const double m1 = min_margin(s.data()), m2 = min_margin(lam.data()); if (m1 < 1.0) add_e(s.data(), 1.0 - m1); if (m2 < 1.0) add_e(lam.data(), 1.0 - m2); }
                    }
                } else {
                    double am_s = 1e300, am_l = 1e300;
                    for (long i = 0; i < ROWS; i++) {
                        const double d = -rz[i] - gd[i];
                        ds[i] = d;
                        if (is_dead(i) || is_soc(i)) continue;
                        if (d < 0.0) am_s = std::min(am_s, -s[i] / d);
                        if (dl[i] < 0.0) am_l = std::min(am_l, -lam[i] / dl[i]);
                    }
                    for (int k = 0; k < N; k++) for (int c = 0; c < nsoc; c++) { const long b0 = (long)k * RS + S::R_SOC + 4 * c; am_s = std::min(am_s, soc_step(&s[b0], &ds[b0])); am_l = std::min(am_l, soc_step(&lam[b0], &dl[b0])); }
                    const double am = std::min(am_s, am_l);
                    if (phase == 0) { const double a_aff = std::min(1.0, am); sigma = (1.0 - a_aff) * (1.0 - a_aff) * (1.0 - a_aff); }
                    else {
                        // experiment (SCP_CPU_GONDZIO = number of correctors): Gondzio's multiple centrality correctors on the linear rows --
                        // aim at a longer step, move the complementarity products of the trial point that left [b_min, b_max] mu_t back to
                        // the box with ONE more solve on the same factorisation (zero residual right-hand sides), keep it if the step grows
                        static const int n_gondzio = std::getenv("SCP_CPU_GONDZIO") ? std::atoi(std::getenv("SCP_CPU_GONDZIO")) : 0;
                        double am_cur = am;
                        for (int gc = 0; gc < n_gondzio && am_cur < 1.0 / opt.step_frac; gc++) {
                            const double a_t = std::min(1.0, opt.step_frac * am_cur * 1.0 + 0.1 + 0.08 * am_cur), mu_t = sigma * mu, bmin = 0.1, bmax = 10.0;
                            std::vector<double> tcor(ROWS, 0.0), zx(XI, 0.0), cxi(XI), cgd(ROWS), cdl(ROWS);
                            for (long i = 0; i < ROWS; i++) {
                                if (is_dead(i) || is_soc(i)) continue;
                                const double v = (s[i] + a_t * ds[i]) * (lam[i] + a_t * dl[i]);
                                double t = v < bmin * mu_t ? bmin * mu_t - v : (v > bmax * mu_t ? bmax * mu_t - v : 0.0);
                                if (t < -bmax * mu_t) t = -bmax * mu_t;
                                tcor[i] = t / lam[i];
                            }
                            newton(w.data(), tcor.data(), zx.data(), cxi.data(), nuv.data());
                            finish(w.data(), tcor.data(), zx.data(), cxi.data(), nuv.data(), cgd.data(), cdl.data());
                            double as2 = 1e300, al2 = 1e300;
                            std::vector<double> ds2(ROWS), dl2(ROWS);
                            for (long i = 0; i < ROWS; i++) {
                                ds2[i] = ds[i] - cgd[i]; dl2[i] = dl[i] + cdl[i];
                                if (is_dead(i) || is_soc(i)) continue;
                                if (ds2[i] < 0.0) as2 = std::min(as2, -s[i] / ds2[i]);
                                if (dl2[i] < 0.0) al2 = std::min(al2, -lam[i] / dl2[i]);
                            }
                            for (int k = 0; k < N; k++) for (int c = 0; c < nsoc; c++) { const long b0 = (long)k * RS + S::R_SOC + 4 * c; as2 = std::min(as2, soc_step(&s[b0], &ds2[b0])); al2 = std::min(al2, soc_step(&lam[b0], &dl2[b0])); }
                            const double am2 = std::min(as2, al2);
                            gondzio_tried++;
                            if (std::min(1.0, opt.step_frac * am2) >= 1.01 * std::min(1.0, opt.step_frac * am_cur)) {
                                gondzio_kept++;
                                for (long i = 0; i < ROWS; i++) { ds[i] = ds2[i]; dl[i] = dl2[i]; gd[i] += cgd[i]; }
                                for (long i = 0; i < XI; i++) dxi[i] += cxi[i];
                                am_cur = am2;
                            } else break;
                        }
                        double alpha = std::min(1.0, opt.step_frac * am_cur);
                        for (int bt = 0; bt < 60; bt++) {
                            for (long i = 0; i < ROWS; i++) { sn[i] = s[i] + alpha * ds[i]; ln[i] = lam[i] + alpha * dl[i]; }
                            for (int r = 0; r < 2 * nx; r++) { ROW(sn.data(), N - 1, r) = 1.0; ROW(ln.data(), N - 1, r) = 1.0; }
                            bool ok = min_margin(sn.data()) > 0.0 && min_margin(ln.data()) > 0.0;
                            if (ok && opt.cgamma > 0.0) {   // experiment: stay in the wide neighbourhood s_i lam_i >= cgamma * mu
                                double g = 0.0, mp = 1e300;
                                for (long i = 0; i < ROWS; i++) { if (is_dead(i)) continue; g += sn[i] * ln[i]; if (!is_soc(i)) mp = std::min(mp, sn[i] * ln[i]); }
                                ok = mp >= opt.cgamma * g / deg;
                            }
                            if (ok) break;
                            alpha *= 0.8;
                        }
                        s.swap(sn); lam.swap(ln); trace_alpha = alpha;
                        for (long i = 0; i < XI; i++) xi[i] += alpha * dxi[i];
                    }
                }
            }
        }
        // ECOS "reduced tolerances" -> ALMOST_OPTIMAL (as the device solver)
        if (res.status != 0 && bestr.pres <= 1e-4 && bestr.dres <= 1e-4 && (bestr.gap <= 5e-5 || bestr.relgap <= 5e-5)) res.status = 1;
        xi_prev = xi; lam_prev = lam;
        // a warm solve that ended before it could refresh a snapshot (0 iterations on a converged reference) keeps the old one
        const bool keep = it0 == 0 && !std::getenv("SCP_CPU_SNAP_NOKEEP");
        if (opt.warm >= 3) snap_ok = snap_taken || (keep && snap_ok);
        if (opt.warm == 4) snapA_ok = snapA_taken || (keep && snapA_ok);
        if (opt.warm == 5) for (int l = 0; l < IpmOpts::NL; l++) { sn_new[l] = sn_taken[l]; sn_ok[l] = sn_taken[l] || ((keep || opt.cross) && it0 == 0 && sn_ok[l]); }
        const int its = res.iters, stt = res.status;
        res = bestr; res.iters = its; res.status = stt;
        return res;
    }
};

// ------------------------------------------------------------------------------------------------
// PTR loop for one problem (src/solvers/ptr.jl:448-532, fixed iteration count: eps_abs = eps_rel = 0)
// ------------------------------------------------------------------------------------------------
struct PtrOut { double J, Jtr, Jvc; int ipm_iters, ipm_status_worst, feas; double t_disc, t_form, t_solve; };

template <class M>
static void ptr_one(const double* par, int N, int Nsub, int iters, double wvc, double wtr, double feas_tol, const double* Sx,
                    const double* cx, const double* Su, const double* cu, const double* Sp, const double* cp, const double* pp,
                    double* xd, double* ud, double* p, PtrOut* out, double* hist /* [iters][6] or null */)
{
    using S = SP<M>;
    constexpr int nx = S::nx, nu = S::nu, np = S::np, npa = S::npa, nz = S::nz, npF = M::npF > 0 ? M::npF : 1;
    const int Mi = N - 1;
    std::vector<double> A((size_t)nx * nx * Mi), Bm((size_t)nx * nu * Mi), Bp((size_t)nx * nu * Mi), F((size_t)nx * npa * Mi), r((size_t)nx * Mi),
        E((size_t)nx * nx * Mi), defect((size_t)nx * Mi), iSx(nx), Fc((size_t)nx * npF * Mi);
    for (int i = 0; i < nx; i++) iSx[i] = 1.0 / Sx[i];
    typename M::Params P = M::make_params(par);
    const typename S::Off o = S::offsets(N);
    std::vector<double> slab(o.total, 0.0), best;
    CpuIpm<M> ipm;
    int feas = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto disc = [&]() {
        // oracle_discretize returns the dense F[nx,np,N-1]; the stage form wants the npF structurally non-zero columns
        oracle_discretize(M::id, par, N, Nsub, xd, ud, p, iSx.data(), feas_tol, A.data(), Bm.data(), Bp.data(), F.data(), r.data(), E.data(),
                          defect.data(), &feas);
        for (int k = 0; k < Mi; k++) for (int jj = 0; jj < M::npF; jj++) for (int i = 0; i < nx; i++) Fc[((size_t)k * npF + jj) * nx + i] = F[((size_t)k * npa + M::Fcol(jj)) * nx + i];
    };
    out->t_disc = out->t_form = out->t_solve = 0; out->ipm_iters = 0; out->ipm_status_worst = 0;
    bool warm_ok = false;
    int last_status = 0;
    double prev_dev = 1e300;
    int cold_iters = 0;
    double t0 = now();
    disc();
    out->t_disc += now() - t0;
    for (int it = 0; it < iters; it++) {
        t0 = now();
        AsmArgs aa;
        aa.B = 1; aa.N = N; aa.wvc = wvc; aa.wtr = wtr; aa.xd = xd; aa.ud = ud; aa.p = p; aa.pp = pp;
        aa.A = A.data(); aa.Bm = Bm.data(); aa.Bp = Bp.data(); aa.F = Fc.data(); aa.r = r.data();
        aa.Sx = Sx; aa.cx = cx; aa.Su = Su; aa.cu = cu; aa.Sp = Sp; aa.cp = cp; aa.slab = slab.data(); aa.slab_stride = o.total; aa.active = nullptr;
        for (int k = 0; k <= N; k++) ptr_assemble_entry<M>(aa, P, 0, k);
        double t1 = now();
        out->t_form += t1 - t0;
        ipm.bind(slab.data(), N);
        if (const char* e = std::getenv("SCP_CPU_WARM")) ipm.opt.warm = std::atoi(e);
        if (const char* e = std::getenv("SCP_CPU_WARM_MU")) ipm.opt.warm_mu = std::atof(e);
        if (const char* e = std::getenv("SCP_CPU_WARM_SAVE_MU")) ipm.opt.warm_save_mu = std::atof(e);
        if (const char* e = std::getenv("SCP_CPU_STEPFRAC")) ipm.opt.step_frac = std::atof(e);
        if (const char* e = std::getenv("SCP_CPU_STALL_REL")) ipm.opt.stall_rel = std::atof(e);
        if (const char* e = std::getenv("SCP_CPU_CGAMMA")) ipm.opt.cgamma = std::atof(e);
        if (const char* e = std::getenv("SCP_CPU_WARM_FROM")) ipm.opt.warm_from = std::atoi(e);
        if (const char* e = std::getenv("SCP_CPU_WARM_MAXIT")) ipm.opt.warm_max_iter = std::atoi(e);
        if (const char* e = std::getenv("SCP_CPU_NREF")) ipm.opt.nref = std::atoi(e);
        if (const char* e = std::getenv("SCP_CPU_REFCORR")) ipm.opt.ref_corrector_only = std::atoi(e);
        if (const char* e = std::getenv("SCP_CPU_REFSTALL")) ipm.opt.ref_on_stall = std::atoi(e);
        if (const char* e = std::getenv("SCP_CPU_REFGAP")) ipm.opt.ref_gap = std::atof(e);
        ipm.opt.reg = 1e-12;   // per solve; escalated x10 by a factorisation that breaks down (device: reg_cur)
        if (const char* e = std::getenv("SCP_CPU_REG")) ipm.opt.reg = std::atof(e);
        if (const char* e = std::getenv("SCP_CPU_REGESC")) ipm.opt.reg_escalate = std::atoi(e);
        double warm_dev = ipm.opt.warm_dev;
        if (const char* e = std::getenv("SCP_CPU_WARM_DEV")) warm_dev = std::atof(e);
        int warm_min_cold = ipm.opt.warm_min_cold;
        if (const char* e = std::getenv("SCP_CPU_WARM_MINCOLD")) warm_min_cold = std::atoi(e);
        ipm.use_warm = ipm.opt.warm > 0 && it >= ipm.opt.warm_from && warm_ok && prev_dev <= warm_dev && cold_iters >= warm_min_cold;
        if (const char* e = std::getenv("SCP_CPU_WARM_SAVE_MU_COARSE")) ipm.opt.warm_save_mu_coarse = std::atof(e);
        if (ipm.opt.warm == 4) {   // two snapshot levels: the fine one for small reference deviations, the coarse one otherwise
            ipm.snap_level = prev_dev <= warm_dev ? 1 : 0;
            ipm.use_warm = it >= ipm.opt.warm_from && warm_ok && (ipm.snap_level == 1 ? ipm.snap_ok : (ipm.snapA_ok && cold_iters >= warm_min_cold));
        }
        if (ipm.opt.warm == 5) {
            if (const char* e = std::getenv("SCP_CPU_LVL_MU")) std::sscanf(e, "%lf,%lf,%lf,%lf", &ipm.opt.lvl_mu[0], &ipm.opt.lvl_mu[1], &ipm.opt.lvl_mu[2], &ipm.opt.lvl_mu[3]);
            if (const char* e = std::getenv("SCP_CPU_LVL_DEV")) std::sscanf(e, "%lf,%lf,%lf,%lf", &ipm.opt.lvl_dev[0], &ipm.opt.lvl_dev[1], &ipm.opt.lvl_dev[2], &ipm.opt.lvl_dev[3]);
            if (const char* e = std::getenv("SCP_CPU_LVL_CAP")) std::sscanf(e, "%d,%d,%d,%d", &ipm.opt.lvl_cap[0], &ipm.opt.lvl_cap[1], &ipm.opt.lvl_cap[2], &ipm.opt.lvl_cap[3]);
            if (const char* e = std::getenv("SCP_CPU_ALMOST_LVL")) ipm.opt.almost_lvl = std::atoi(e);
            if (const char* e = std::getenv("SCP_CPU_CROSS")) ipm.opt.cross = std::atoi(e);
            if (const char* e = std::getenv("SCP_CPU_LVL_FLOOR")) ipm.opt.lvl_floor = std::atof(e);
            for (int l = 0; l < IpmOpts::NL; l++) ipm.sn_acc[l] = (ipm.sn_new[l] ? 0.0 : ipm.sn_acc[l]) + prev_dev;
            for (int l = 0; l < IpmOpts::NL; l++) ipm.sn_ok_prev[l] = ipm.sn_ok[l];
            int lvl = -1;
            for (int l = (last_status == 1 ? ipm.opt.almost_lvl : IpmOpts::NL - 1); l >= 0; l--)
                if ((ipm.opt.cross ? ipm.sn_acc[l] : prev_dev) <= ipm.opt.lvl_dev[l] && ipm.sn_ok[l] && (l > 0 || cold_iters >= warm_min_cold)) { lvl = l; break; }
            ipm.snap_level = lvl < 0 ? 0 : lvl;
            ipm.use_warm = it >= ipm.opt.warm_from && warm_ok && lvl >= 0;
        }
        if (std::getenv("SCP_CPU_COLD_STRUCT")) { ipm.opt.warm = 1; ipm.use_warm = true; }
        const bool was_warm = ipm.use_warm;
        IpmResult rr = ipm.solve(best);
        // warm start failed, or ended at reduced accuracy with a primal / dual residual above the tolerance (a cold
        // ALMOST_OPTIMAL exit always has residuals at round-off: only the gap stalls): cold restart, iterations of both counted
        auto warm_failed = [&](const IpmResult& r_) { return r_.status > 1 || (r_.status == 1 && (r_.pres > ipm.opt.feastol || r_.dres > ipm.opt.feastol)); };
        // round 6: a failed start from the very fine level is repeated from the next level (its snapshot is still there: a start at
        // mu ~ 1e-10 never produces an iterate within two decades of 1e-8) before the cold repeat -- SCP_CPU_FALLBACK=0 goes cold at once
        if (ipm.opt.warm == 5 && ipm.use_warm && ipm.snap_level == IpmOpts::NL - 1 && ipm.sn_ok_prev[IpmOpts::NL - 2] && warm_failed(rr) &&
            !(std::getenv("SCP_CPU_FALLBACK") && std::atoi(std::getenv("SCP_CPU_FALLBACK")) == 0)) {
            const int it_w = rr.iters;
            ipm.snap_level = IpmOpts::NL - 2;
            ipm.sn_ok[IpmOpts::NL - 2] = true;
            rr = ipm.solve(best);
            rr.iters += it_w;
        }
        if (ipm.use_warm && (rr.status > 1 || (rr.status == 1 && (rr.pres > ipm.opt.feastol || rr.dres > ipm.opt.feastol)))) {
            const int it_w = rr.iters;
            ipm.use_warm = false;
            rr = ipm.solve(best);
            rr.iters += it_w;
        }
        if (rr.status > 1) {   // cold solve failed: once more with the refinement on from the first iteration (device: attempt 2)
            const int it_c = rr.iters;
            const double rg = ipm.opt.ref_gap;
            const int nr = ipm.opt.nref;
            ipm.opt.ref_gap = 1e300; ipm.opt.nref = nr > 0 ? nr : 1;
            ipm.use_warm = false;
            rr = ipm.solve(best);
            rr.iters += it_c;
            ipm.opt.ref_gap = rg; ipm.opt.nref = nr;
        }
        if (std::getenv("SCP_CPU_ATTEMPTS"))     // diagnostic: which attempts a solve went through
            std::fprintf(stderr, "ATT it %d warm %d level %d dev %.3e iters %d status %d snap %d%d\n", it, (int)was_warm, ipm.snap_level, prev_dev, rr.iters,
                         rr.status, (int)ipm.snapA_ok, (int)ipm.snap_ok);
        warm_ok = rr.status <= 1;
        last_status = rr.status;
        if (!was_warm) cold_iters = rr.iters;   // iterations of the last COLD solve: warm starts pay only where cold solves are slow
        {   // deviation of this solution from its reference (scaled, inf-norm): solution_deviation, scp.jl:909-931 (q = Inf)
            double dx = 0.0, dpv = 0.0;
            for (int k = 0; k < N; k++) for (int j = 0; j < nx; j++) dx = std::max(dx, std::fabs(best[(size_t)k * nz + j] - slab[(size_t)k * S::SR + S::O_ZREF + j]));
            for (int j = 0; j < np; j++) dpv = std::max(dpv, std::fabs(best[(size_t)N * (nz + S::AS) + j] - slab[o.pref + j]));
            prev_dev = dx + dpv;
        }
        double t2 = now();
        out->t_solve += t2 - t1;
        out->ipm_iters += rr.iters; out->ipm_status_worst = std::max(out->ipm_status_worst, rr.status);
        // un-scale (value(blk), src/parser/block.jl:368-394)
        for (int k = 0; k < N; k++) {
            for (int i = 0; i < nx; i++) xd[(size_t)k * nx + i] = Sx[i] * best[(size_t)k * nz + i] + cx[i];
            for (int i = 0; i < nu; i++) ud[(size_t)k * nu + i] = Su[i] * best[(size_t)k * nz + nx + i] + cu[i];
        }
        for (int j = 0; j < np; j++) p[j] = Sp[j] * best[(size_t)N * (nz + S::AS) + j] + cp[j];
        if (hist) { hist[it * 6 + 0] = rr.pcost; hist[it * 6 + 1] = rr.gap; hist[it * 6 + 2] = rr.pres; hist[it * 6 + 3] = rr.dres; hist[it * 6 + 4] = rr.iters; hist[it * 6 + 5] = rr.status; }
        if (rr.status > 1) break;   // unsafe solution (scp.jl:965-980)
        disc();
        out->t_disc += now() - t2;
    }
    out->feas = feas;
}

template <class Fn>
static int with_model(int model_id, Fn&& fn)
{
    switch (model_id) {
        case 0: return fn(DoubleIntegrator{});
        case 1: return fn(Quadrotor{});
        case 2: return fn(RocketLanding{});
        default: return 2;
    }
}
}  // namespace

// Batched PTR solve on the host: arrays in the C-ABI layout of include/scp_mi355x.h (trailing batch dimension), guesses in
// xd/ud/p on entry, solutions on exit.  threads <= 0: all cores.  stats[B][8] = (ipm iterations, worst ipm status, feas,
// t_discretize, t_formulate, t_solve, -, -); hist may be NULL or [B][iters][6] = (pcost, gap, pres, dres, ipm iters, status).
// deadline_s > 0: problems are not STARTED after that many seconds (bounded sample for bench.py); stats[b][7] = 1 marks the
// problems that ran, *n_done their count.
extern "C" int cpu_ptr_solve_batch(int model_id, const double* par, int N, int Nsub, int iters, double wvc, double wtr, double feas_tol,
                                   const double* Sx, const double* cx, const double* Su, const double* cu, const double* Sp,
                                   const double* cp, int B, const double* pp, double* xd, double* ud, double* p, int threads,
                                   double* stats, double* hist, double* seconds, double deadline_s, int* n_done)
{
    if (threads > 0) omp_set_num_threads(threads);
    int done = 0;
    int rc = with_model(model_id, [&](auto m) -> int {
        using M = decltype(m);
        constexpr int nx = M::nx, nu = M::nu, np = M::np, npp = M::npp;
        const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : done)
        for (int b = 0; b < B; b++) {
            if (deadline_s > 0.0 && omp_get_wtime() - t0 > deadline_s) continue;
            done += 1;
            PtrOut o;
            ptr_one<M>(par, N, Nsub, iters, wvc, wtr, feas_tol, Sx, cx, Su, cu, Sp, cp, pp + (size_t)b * npp, xd + (size_t)b * N * nx,
                       ud + (size_t)b * N * nu, p + (size_t)b * (np > 0 ? np : 0), &o, hist ? hist + (size_t)b * iters * 6 : nullptr);
            if (stats) { double* s = stats + (size_t)b * 8; s[0] = o.ipm_iters; s[1] = o.ipm_status_worst; s[2] = o.feas; s[3] = o.t_disc; s[4] = o.t_form; s[5] = o.t_solve; s[6] = 0; s[7] = 1; }
        }
        if (seconds) *seconds = omp_get_wtime() - t0;
        return 0;
    });
    if (n_done) *n_done = done;
    return rc;
}

extern "C" int cpu_ptr_max_threads() { return omp_get_num_procs(); }
