// K3: batched stage-structured primal-dual interior-point solver for the reduced PTR subproblem
// (replaces JuMP `optimize!` -> ECOS, src/parser/program.jl:419-424 / src/solvers/scp.jl:942-950).
//
// One WAVEFRONT (64 lanes, one 64-thread workgroup) per problem; the whole Mehrotra
// predictor-corrector iteration runs inside one launch, so a Monte-Carlo batch is one grid.
// Algorithm = oracle/ipm_struct.py (numpy mirror used by the tests), which is oracle/ipm.py
// (ECOS-class NT-scaled primal-dual method) specialised to the stage structure:
//   * every constraint is an inequality / second-order cone row; epigraph variables of the
//     L1 / L_inf / hinge penalties are eliminated analytically from the Newton system;
//   * penalised equality-like rows (dynamics, boundary conditions, hinge) stay in AUGMENTED
//     form (unknown nu): the Newton matrix is the quasi-definite interleaved chain
//         z_0 - nu_0 - z_1 - nu_1 - ... - z_{N-1} - nu_{N-1}
//     factorised by a forward sweep in which every Schur complement ADDS positive
//     semidefinite terms (no cancellation; 1/kappa enters only as a tiny regulariser):
//         Sz_k  = H0_k + X_{k-1}' X_{k-1},          X_k = Lnu_k^-1 Et_k
//         Snu_k = diag(1/kappa_k + reg) + Y_k' Y_k, Y_k = Lz_k^-1 Dt_k'
//   * the parameter block p is an arrow handled with np extra right-hand sides;
//   * multipliers of penalised rows are recovered from nu and the aux dual-feasibility
//     rows, never as w*(G dxi + r) with w = lam/s ~ omega^2/mu.
// Small dense blocks (<= 16x16, fp64) live in LDS; lanes split matrix entries / columns.
// These products are far too small for MFMA tiles (v_mfma_f64_16x16x4 needs 16x16 outputs
// with K>=4 from one wave; the blocks here are 11x11..16x16 but latency-bound chains), see
// DESIGN.md for the measured discussion.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage_problem.hpp"

namespace scp {

struct IpmArgs {
    int B, N;
    int max_iter, nref, stall;
    double feastol, abstol, reltol, reg, ref_gap, ref_tol;
    const double* slab;  // problem data [B][slab_stride]
    long slab_stride;
    double* work;  // workspace [B][work_stride]
    long work_stride;
    // outputs
    double* z_out;    // [B][N*nz] scaled primal
    double* p_out;    // [B][npa]
    int* status;      // [B]: 0 OPTIMAL, 1 ALMOST_OPTIMAL, 2 ITERATION_LIMIT, 3 NUMERICAL_ERROR
    int* iters;       // [B]
    double* info;     // [B][8]: pcost(+const), dcost, gap, pres, dres, relgap, merit, best_it
    const int* active;  // optional [B]: problems with active[b] == 0 are skipped
    long long* prof;    // optional [B][8] phase counters (100 MHz wall clock ticks)
};

enum { IPM_OPTIMAL = 0, IPM_ALMOST = 1, IPM_ITERLIM = 2, IPM_NUMERR = 3 };

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

template <class M>
struct IpmWork {
    using S = SP<M>;
    // xi-vector: [z N*nz | aux N*AS | p npa | gaux AG]
    __host__ __device__ static long XI(int N) { return (long)N * (S::nz + S::AS) + S::npa + S::AG; }
    // row-vector: [rows N*RS | grows RG]
    __host__ __device__ static long ROWS(int N) { return (long)N * S::RS + S::RG; }
    struct Off {
        long xi, dxi, rx, exi, best, rxe;                       // xi-vectors
        long s, lam, rz, w, rtil, ds, dl, gd, r2, el, hneg, ge;  // row-vectors
        long socW;                                              // [N][nsoc][36]: W(16) Wi(16) lamt(4)
        long Lz, Lnu, X, Y, C0, Ycz, Ycnu, fb, ft, nuv, kinv;  // newton
        long total;
    };
    __host__ __device__ static Off offsets(int N)
    {
        Off o;
        long c = 0;
        auto take = [&](long n) { long r = c; c += (n + 1) & ~1L; return r; };
        const long xi = XI(N), rows = ROWS(N);
        o.xi = take(xi); o.dxi = take(xi); o.rx = take(xi); o.exi = take(xi); o.best = take(xi); o.rxe = take(xi);
        o.s = take(rows); o.lam = take(rows); o.rz = take(rows); o.w = take(rows); o.rtil = take(rows);
        o.ds = take(rows); o.dl = take(rows); o.gd = take(rows); o.r2 = take(rows); o.el = take(rows);
        o.hneg = take(rows); o.ge = take(rows);
        o.socW = take((long)N * (S::nsoc > 0 ? S::nsoc : 1) * 36);
        o.Lz = take((long)N * S::nz * S::nz); o.Lnu = take((long)N * S::MNU * S::MNU);
        o.X = take((long)N * S::MNU * S::nz); o.Y = take((long)N * S::nz * S::MNU);
        o.C0 = take((long)N * S::nz * S::npa);
        o.Ycz = take((long)N * S::nz * S::npa); o.Ycnu = take((long)N * S::MNU * S::npa);
        o.fb = take((long)N * S::nz * (1 + S::npa)); o.ft = take((long)N * S::MNU * (1 + S::npa));
        o.nuv = take((long)N * S::MNU); o.kinv = take((long)N * S::MNU);
        o.total = (c + 7) & ~7L;
        return o;
    }
};

template <class M>
struct Ipm {
    using S = SP<M>;
    using WK = IpmWork<M>;
    static constexpr int nx = S::nx, nu = S::nu, np = S::np, npa = S::npa, nz = S::nz, ns = S::ns, nl = S::nl,
                         nsoc = S::nsoc, ml = S::ml, ng = S::ng, nic = S::nic, ntc = S::ntc, nbc = S::nbc, RS = S::RS,
                         RG = S::RG, AS = S::AS, AG = S::AG, MNU = S::MNU;
    static constexpr int NR = 1 + npa;  // right-hand sides carried through the factor sweep
    // LDS scratch (one wave per block)
    struct Lds {
        double Sz[nz * nz];
        double Snu[MNU * MNU];
        double X[MNU * nz];
        double Y[nz * MNU];
        double Ysoc[(nsoc > 0 ? 4 * nsoc : 1) * nz];
        double Cz[nz * npa];
        double rb[nz * NR];   // b-hat columns
        double rt[MNU * NR];  // t-hat columns
        double tmp[64];
        int fail;
    };

    int N, lane;
    const double* P;
    typename S::Off o;
    typename WK::Off wo;
    double* W;
    Lds* L;
    IpmArgs a;
    double ttrp, cost_const;
    // optional phase cycle counters (IpmArgs.prof != nullptr): 0 G_apply, 1 GT_apply, 2 factor, 3 solve fwd+rhs,
    // 4 solve backward, 5 aux recovery + dlam, 6 step length / update / residual sums
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    __device__ long long tick() const { return (long long)wall_clock64(); }

    // ---------------- accessors ----------------
    __device__ double& Z(double* v, int k, int j) const { return v[(long)k * nz + j]; }
    __device__ double& AUX(double* v, int k, int i) const { return v[(long)N * nz + (long)k * AS + i]; }
    __device__ double& PV(double* v, int j) const { return v[(long)N * (nz + AS) + j]; }
    __device__ double& GAUX(double* v, int i) const { return v[(long)N * (nz + AS) + npa + i]; }
    __device__ double& ROW(double* v, int k, int r) const { return v[(long)k * RS + r]; }
    __device__ double& GROW(double* v, int r) const { return v[(long)N * RS + r]; }
    __device__ const double* Dk(int k) const { return P + o.D(k); }
    __device__ const double* Ek(int k) const { return P + o.E(k); }
    __device__ const double* Fpk(int k) const { return P + o.Fp(k); }
    __device__ const double* Klk(int k) const { return P + o.Kl(k); }
    __device__ const double* Kpk(int k) const { return P + o.Kp(k); }
    __device__ void sync() const { __syncthreads(); }

    // is row r of stage k a live row (dyn rows do not exist at the last node)?
    __device__ bool live(int k, int r) const { return !(k == N - 1 && r < 2 * nx); }

    // ---------------- linear part of the row functions: out = G * v ----------------
    __device__ void G_apply(double* v, double* out)
    {
        const long long t0_ = tick();
        for (int k = 0; k < N; k++) {
            for (int r = lane; r < RS; r += 64) {
                double val = 0.0;
                if (r < 2 * nx) {
                    const int i = r % nx;
                    if (k < N - 1) {
                        double acc = 0.0;
                        const double *d = Dk(k) + i * nz, *e = Ek(k) + i * nz, *f = Fpk(k) + i * npa;
                        for (int j = 0; j < nz; j++) acc += d[j] * Z(v, k, j) + e[j] * Z(v, k + 1, j);
                        for (int j = 0; j < np; j++) acc += f[j] * PV(v, j);
                        val = (r < nx ? acc : -acc) - AUX(v, k, S::A_Y + i);
                    }
                } else if (r < S::R_TR0) {
                    const int i = (r - S::R_H0) % (ns > 0 ? ns : 1);
                    if (r < S::R_H1) {
                        double acc = 0.0;
                        const double *g = Klk(k) + i * nz, *gp = Kpk(k) + i * npa;
                        for (int j = 0; j < nz; j++) acc += g[j] * Z(v, k, j);
                        for (int j = 0; j < np; j++) acc += gp[j] * PV(v, j);
                        val = acc - AUX(v, k, S::A_V + i);
                    } else {
                        val = -AUX(v, k, S::A_V + i);
                    }
                } else if (r < S::R_LIN) {
                    const int j = (r - S::R_TR0) % nz;
                    const double eta = AUX(v, k, j < nx ? S::A_EX : S::A_EU);
                    val = (r < S::R_TR1 ? Z(v, k, j) : -Z(v, k, j)) - eta;
                } else {
                    const int row = ns + (r - S::R_LIN);  // lin rows then soc rows are contiguous in Kl
                    double acc = 0.0;
                    const double *g = Klk(k) + row * nz, *gp = Kpk(k) + row * npa;
                    for (int j = 0; j < nz; j++) acc += g[j] * Z(v, k, j);
                    for (int j = 0; j < np; j++) acc += gp[j] * PV(v, j);
                    val = (r < S::R_SOC) ? acc : -acc;  // cone rows: G = -M
                }
                ROW(out, k, r) = val;
            }
        }
        for (int r = lane; r < RG; r += 64) {
            double val = 0.0;
            if (r < S::G_TC0) {
                const int i = r % nic;
                double acc = 0.0;
                for (int j = 0; j < nx; j++) acc += P[o.H0 + i * nx + j] * Z(v, 0, j);
                for (int j = 0; j < np; j++) acc += P[o.K0 + i * npa + j] * PV(v, j);
                val = (r < S::G_IC1 ? acc : -acc) - GAUX(v, S::GA_YIC + i);
            } else if (r < S::G_TRP0) {
                const int i = (r - S::G_TC0) % ntc;
                double acc = 0.0;
                for (int j = 0; j < nx; j++) acc += P[o.Hf + i * nx + j] * Z(v, N - 1, j);
                for (int j = 0; j < np; j++) acc += P[o.Kf + i * npa + j] * PV(v, j);
                val = (r < S::G_TC1 ? acc : -acc) - GAUX(v, S::GA_YTC + i);
            } else if (r < S::G_LIN) {
                const int j = (r - S::G_TRP0) % (np > 0 ? np : 1);
                val = (r < S::G_TRP1 ? PV(v, j) : -PV(v, j)) - GAUX(v, S::GA_EP);
            } else {
                const int i = r - S::G_LIN;
                double acc = 0.0;
                for (int j = 0; j < np; j++) acc += P[o.Lp + i * npa + j] * PV(v, j);
                val = acc;
            }
            GROW(out, r) = val;
        }
        sync();
        prof[0] += tick() - t0_;
    }

    // ---------------- out = G' * mu ----------------
    __device__ void GT_apply(double* mu, double* out)
    {
        const long long t0_ = tick();
        double pacc[npa];
        for (int j = 0; j < npa; j++) pacc[j] = 0.0;
        for (int k = 0; k < N; k++) {
            if (lane < nz) {
                const int j = lane;
                double acc = 0.0;
                if (k < N - 1) {
                    const double* d = Dk(k);
                    for (int i = 0; i < nx; i++) acc += d[i * nz + j] * (ROW(mu, k, i) - ROW(mu, k, nx + i));
                }
                if (k > 0) {
                    const double* e = Ek(k - 1);
                    for (int i = 0; i < nx; i++) acc += e[i * nz + j] * (ROW(mu, k - 1, i) - ROW(mu, k - 1, nx + i));
                }
                acc += ROW(mu, k, S::R_TR0 + j) - ROW(mu, k, S::R_TR1 + j);
                const double* g = Klk(k);
                for (int i = 0; i < ns; i++) acc += g[i * nz + j] * ROW(mu, k, S::R_H0 + i);
                for (int i = 0; i < nl; i++) acc += g[(ns + i) * nz + j] * ROW(mu, k, S::R_LIN + i);
                for (int i = 0; i < 4 * nsoc; i++) acc -= g[(ns + nl + i) * nz + j] * ROW(mu, k, S::R_SOC + i);
                if (j < nx) {
                    if (k == 0)
                        for (int i = 0; i < nic; i++) acc += P[o.H0 + i * nx + j] * (GROW(mu, S::G_IC0 + i) - GROW(mu, S::G_IC1 + i));
                    if (k == N - 1)
                        for (int i = 0; i < ntc; i++) acc += P[o.Hf + i * nx + j] * (GROW(mu, S::G_TC0 + i) - GROW(mu, S::G_TC1 + i));
                }
                Z(out, k, j) = acc;
            } else if (lane < nz + AS) {
                const int i = lane - nz;
                double acc = 0.0;
                if (i < nx) acc = (k < N - 1) ? -(ROW(mu, k, i) + ROW(mu, k, nx + i)) : 0.0;
                else if (i < nx + ns) acc = -(ROW(mu, k, S::R_H0 + i - nx) + ROW(mu, k, S::R_H1 + i - nx));
                else if (i == S::A_EX) { for (int j = 0; j < nx; j++) acc -= ROW(mu, k, S::R_TR0 + j) + ROW(mu, k, S::R_TR1 + j); }
                else { for (int j = nx; j < nz; j++) acc -= ROW(mu, k, S::R_TR0 + j) + ROW(mu, k, S::R_TR1 + j); }
                AUX(out, k, i) = acc;
            }
            if (np > 0) {
                // p contributions of this stage, spread over lanes by row then reduced at the end
                for (int r = lane; r < nx + ml; r += 64) {
                    if (r < nx) {
                        if (k < N - 1) {
                            const double m = ROW(mu, k, r) - ROW(mu, k, nx + r);
                            for (int j = 0; j < np; j++) pacc[j] += Fpk(k)[r * npa + j] * m;
                        }
                    } else {
                        const int row = r - nx;
                        double m;
                        if (row < ns) m = ROW(mu, k, S::R_H0 + row);
                        else if (row < ns + nl) m = ROW(mu, k, S::R_LIN + row - ns);
                        else m = -ROW(mu, k, S::R_SOC + row - ns - nl);
                        for (int j = 0; j < np; j++) pacc[j] += Kpk(k)[row * npa + j] * m;
                    }
                }
            }
        }
        if (np > 0) {
            for (int j = 0; j < np; j++) {
                double t = wave_sum(pacc[j]);
                if (lane == 0) {
                    t += GROW(mu, S::G_TRP0 + j) - GROW(mu, S::G_TRP1 + j);
                    for (int i = 0; i < ng; i++) t += P[o.Lp + i * npa + j] * GROW(mu, S::G_LIN + i);
                    for (int i = 0; i < nic; i++) t += P[o.K0 + i * npa + j] * (GROW(mu, S::G_IC0 + i) - GROW(mu, S::G_IC1 + i));
                    for (int i = 0; i < ntc; i++) t += P[o.Kf + i * npa + j] * (GROW(mu, S::G_TC0 + i) - GROW(mu, S::G_TC1 + i));
                    PV(out, j) = t;
                }
            }
        } else if (lane == 0) {
            PV(out, 0) = 0.0;
        }
        for (int i = lane; i < AG; i += 64) {
            double acc = 0.0;
            if (i < nic) acc = -(GROW(mu, S::G_IC0 + i) + GROW(mu, S::G_IC1 + i));
            else if (i < nic + ntc) acc = -(GROW(mu, S::G_TC0 + i - nic) + GROW(mu, S::G_TC1 + i - nic));
            else { for (int j = 0; j < np; j++) acc -= GROW(mu, S::G_TRP0 + j) + GROW(mu, S::G_TRP1 + j); }
            GAUX(out, i) = acc;
        }
        sync();
        prof[1] += tick() - t0_;
    }

    // ---------------- constants: hneg = -h (value added to G xi to get G xi - h) ----------------
    __device__ void build_hneg(double* hn) const
    {
        for (int k = 0; k < N; k++)
            for (int r = lane; r < RS; r += 64) {
                double c = 0.0;
                if (r < 2 * nx) { if (k < N - 1) { const double v = P[o.cd(k) + r % nx]; c = r < nx ? v : -v; } }
                else if (r < S::R_H1) c = P[o.cl(k) + (r - S::R_H0)];
                else if (r < S::R_TR0) c = 0.0;
                else if (r < S::R_LIN) { const int j = (r - S::R_TR0) % nz; const double v = P[o.zref(k) + j]; c = r < S::R_TR1 ? -v : v; }
                else if (r < S::R_SOC) c = P[o.cl(k) + ns + (r - S::R_LIN)];
                else c = -P[o.cl(k) + ns + nl + (r - S::R_SOC)];
                ROW(hn, k, r) = c;
            }
        for (int r = lane; r < RG; r += 64) {
            double c;
            if (r < S::G_TC0) { const double v = P[o.l0 + r % nic]; c = r < S::G_IC1 ? v : -v; }
            else if (r < S::G_TRP0) { const double v = P[o.lf + (r - S::G_TC0) % ntc]; c = r < S::G_TC1 ? v : -v; }
            else if (r < S::G_LIN) { const double v = P[o.pref + (r - S::G_TRP0) % (np > 0 ? np : 1)]; c = r < S::G_TRP1 ? -v : v; }
            else c = P[o.lp + r - S::G_LIN];
            GROW(hn, r) = c;
        }
        sync();
    }

    // cost vector c on the xi layout (main: q, qp; aux: om, hw, ttr, ttr; gaux: bw0, bwf, ttrp)
    __device__ double cvec(int idx_kind, int k, int i) const
    {
        // kind 0: z, 1: aux, 2: p, 3: gaux
        if (idx_kind == 0) return P[o.q(k) + i];
        if (idx_kind == 1) {
            if (i < nx) return k < N - 1 ? P[o.om(k) + i] : 0.0;
            if (i < nx + ns) return P[o.hw(k) + (i - nx)];
            return P[o.ttr(k)];
        }
        if (idx_kind == 2) return P[o.qp + i];
        if (i < nic) return P[o.bw0 + i];
        if (i < nic + ntc) return P[o.bwf + i - nic];
        return np > 0 ? ttrp : 0.0;
    }

    // ---------------- small dense helpers on LDS (row-major, leading dimension ld) ----------------
    __device__ void chol(double* A, int n, int ld) const
    {
        for (int j = 0; j < n; j++) {
            if (lane == 0) {
                const double d = A[j * ld + j];
                if (!(d > 0.0)) L->fail = 1;
                A[j * ld + j] = sqrt(d > 0.0 ? d : 1.0);
            }
            sync();
            const double djj = A[j * ld + j];
            for (int i = j + 1 + lane; i < n; i += 64) A[i * ld + j] /= djj;
            sync();
            const int m = n - j - 1;
            for (int idx = lane; idx < m * m; idx += 64) {
                const int i = j + 1 + idx / m, c = j + 1 + idx % m;
                if (c <= i) A[i * ld + c] -= A[i * ld + j] * A[c * ld + j];
            }
            sync();
        }
    }
    // solve L X = B in place; B is n x m row-major (ldb); lanes over the m columns
    __device__ void trsm_lower(const double* Lm, int n, int ld, double* Bm, int m, int ldb) const
    {
        for (int c = lane; c < m; c += 64)
            for (int i = 0; i < n; i++) {
                double acc = Bm[i * ldb + c];
                for (int j = 0; j < i; j++) acc -= Lm[i * ld + j] * Bm[j * ldb + c];
                Bm[i * ldb + c] = acc / Lm[i * ld + i];
            }
        sync();
    }
    // solve L' X = B in place
    __device__ void trsm_lowerT(const double* Lm, int n, int ld, double* Bm, int m, int ldb) const
    {
        for (int c = lane; c < m; c += 64)
            for (int i = n - 1; i >= 0; i--) {
                double acc = Bm[i * ldb + c];
                for (int j = i + 1; j < n; j++) acc -= Lm[j * ld + i] * Bm[j * ldb + c];
                Bm[i * ldb + c] = acc / Lm[i * ld + i];
            }
        sync();
    }

    // elimination coefficients of a penalised pair (type A: both rows carry +-a; hinge: row 2 is -v only)
    struct Pair {
        double kap, tau, rth, Wt;
    };
    __device__ static Pair pairA(double w1, double w2, double rt1, double rt2, double rxa)
    {
        Pair p;
        const double r1 = w1 * rt1, r2 = w2 * rt2;
        p.Wt = w1 + w2; p.rth = -rxa + r1 + r2;
        p.kap = 4.0 * w1 * w2 / p.Wt;
        p.tau = -(r1 - r2) + (w1 - w2) * p.rth / p.Wt;
        return p;
    }
    __device__ static Pair pairC(double w1, double w2, double rt1, double rt2, double rxa)
    {
        Pair p;
        const double r1 = w1 * rt1, r2 = w2 * rt2;
        p.Wt = w1 + w2; p.rth = -rxa + r1 + r2;
        p.kap = w1 * w2 / p.Wt;
        p.tau = -r1 + w1 * p.rth / p.Wt;
        return p;
    }

    // rows of the nu block of stage k: c in [0,nx) dyn, [nx,nx+ns) hinge, [nx+ns, MNU) bc
    __device__ bool nu_live(int k, int c) const
    {
        if (c < nx) return k < N - 1;
        if (c < nx + ns) return true;
        const int i = c - nx - ns;
        return (k == 0 && i < nic) || (k == N - 1 && i < ntc);
    }
    __device__ int mnu(int k) const { return (k == 0 || k == N - 1) ? MNU : S::MNU_MID; }
    // Dt_k[c][j], Et_k[c][j], Ft_k[c][j]
    __device__ double Dt(int k, int c, int j) const
    {
        if (c < nx) return k < N - 1 ? Dk(k)[c * nz + j] : 0.0;
        if (c < nx + ns) return Klk(k)[(c - nx) * nz + j];
        const int i = c - nx - ns;
        if (j >= nx) return 0.0;
        if (k == 0 && i < nic) return P[o.H0 + i * nx + j];
        if (k == N - 1 && i < ntc) return P[o.Hf + i * nx + j];
        return 0.0;
    }
    __device__ double Ft(int k, int c, int j) const
    {
        if (c < nx) return k < N - 1 ? Fpk(k)[c * npa + j] : 0.0;
        if (c < nx + ns) return Kpk(k)[(c - nx) * npa + j];
        const int i = c - nx - ns;
        if (k == 0 && i < nic) return P[o.K0 + i * npa + j];
        if (k == N - 1 && i < ntc) return P[o.Kf + i * npa + j];
        return 0.0;
    }
    // (w1, w2, rtil1, rtil2, rxaux) of nu-row c of stage k
    __device__ void nu_row_data(int k, int c, const double* w, const double* rt, double* rxv, double& w1, double& w2,
                                double& t1, double& t2, double& rxa, bool& hinge) const
    {
        double* wv = const_cast<double*>(w);
        double* rv = const_cast<double*>(rt);
        hinge = false;
        if (c < nx) { w1 = ROW(wv, k, c); w2 = ROW(wv, k, nx + c); t1 = ROW(rv, k, c); t2 = ROW(rv, k, nx + c); rxa = AUX(rxv, k, S::A_Y + c); }
        else if (c < nx + ns) { const int i = c - nx; hinge = true; w1 = ROW(wv, k, S::R_H0 + i); w2 = ROW(wv, k, S::R_H1 + i); t1 = ROW(rv, k, S::R_H0 + i); t2 = ROW(rv, k, S::R_H1 + i); rxa = AUX(rxv, k, S::A_V + i); }
        else {
            const int i = c - nx - ns;
            if (k == 0) { w1 = GROW(wv, S::G_IC0 + i); w2 = GROW(wv, S::G_IC1 + i); t1 = GROW(rv, S::G_IC0 + i); t2 = GROW(rv, S::G_IC1 + i); rxa = GAUX(rxv, S::GA_YIC + i); }
            else { w1 = GROW(wv, S::G_TC0 + i); w2 = GROW(wv, S::G_TC1 + i); t1 = GROW(rv, S::G_TC0 + i); t2 = GROW(rv, S::G_TC1 + i); rxa = GAUX(rxv, S::GA_YTC + i); }
        }
    }

    // type-B (L_inf block) Schur complement entry (a,b) of a block with pair weights w1[j], w2[j], j in [j0, j0+n)
    __device__ double typeB_entry(const double* w1, const double* w2, int n, int a_, int b_) const
    {
        double Wt = 0.0;
        for (int j = 0; j < n; j++) Wt += w1[j] + w2[j];
        const double ha = w1[a_] - w2[a_];
        if (a_ != b_) return -ha * (w1[b_] - w2[b_]) / Wt;
        double rest = 0.0;
        for (int j = 0; j < n; j++) if (j != a_) rest += w1[j] + w2[j];
        const double d = w1[a_] + w2[a_];
        return 4.0 * w1[a_] * w2[a_] / d + ha * ha * rest / (d * Wt);
    }

    // ---------------- Newton factorisation for the current scalings ----------------
    // w: LP weights (lam/s) row-vector; socW: NT data.  Also forward-substitutes the np arrow
    // columns (C0 / Ft) and forms the np x np Schur complement Sp (its Cholesky is kept in spL).
    double spL[npa * npa];
    __device__ void factor(double* w)
    {
        const long long t0_ = tick();
        double* Lz = W + wo.Lz; double* Lnu = W + wo.Lnu; double* Xg = W + wo.X; double* Yg = W + wo.Y;
        double* C0 = W + wo.C0; double* Ycz = W + wo.Ycz; double* Ycnu = W + wo.Ycnu; double* kinvg = W + wo.kinv;
        double* socW = W + wo.socW;
        double Dp[npa * npa];
        for (int i = 0; i < npa * npa; i++) Dp[i] = 0.0;   // accumulated by lane 0 only
        for (int k = 0; k < N; k++) {
            const double* Kl = Klk(k);
            const double* Kp = Kpk(k);
            // ---- cone rows scaled by W^-1 (factored form, avoids forming W^-2) ----
            for (int idx = lane; idx < 4 * nsoc * nz; idx += 64) {
                const int r = idx / nz, j = idx % nz, c = r / 4, rr = r % 4;
                const double* Wi = socW + ((long)k * nsoc + c) * 36 + 16;
                double acc = 0.0;
                for (int q = 0; q < 4; q++) acc += Wi[rr * 4 + q] * Kl[(ns + nl + 4 * c + q) * nz + j];
                L->Ysoc[r * nz + j] = acc;
            }
            sync();
            // ---- H0_k: Qd + trust region (type B) + lin rows + cone rows ----
            for (int idx = lane; idx < nz * nz; idx += 64) {
                const int a_ = idx / nz, b_ = idx % nz;
                double acc = (a_ == b_) ? P[o.Qd(k) + a_] : 0.0;
                const bool ax = a_ < nx, bx = b_ < nx;
                if (ax == bx) {
                    const int j0 = ax ? 0 : nx, n = ax ? nx : nu;
                    acc += typeB_entry(&ROW(w, k, S::R_TR0 + j0), &ROW(w, k, S::R_TR1 + j0), n, a_ - j0, b_ - j0);
                }
                for (int i = 0; i < nl; i++) acc += ROW(w, k, S::R_LIN + i) * Kl[(ns + i) * nz + a_] * Kl[(ns + i) * nz + b_];
                for (int r = 0; r < 4 * nsoc; r++) acc += L->Ysoc[r * nz + a_] * L->Ysoc[r * nz + b_];
                if (k > 0) {
                    const int m = mnu(k - 1);
                    for (int r = 0; r < m; r++) acc += L->X[r * nz + a_] * L->X[r * nz + b_];
                }
                L->Sz[idx] = acc;
            }
            // ---- C0_k = sum_lin w Kl' Kp (+ cones; their Kp is zero in all current models) ----
            for (int idx = lane; idx < nz * npa; idx += 64) {
                const int a_ = idx / npa, j = idx % npa;
                double acc = 0.0;
                if (np > 0)
                    for (int i = 0; i < nl; i++) acc += ROW(w, k, S::R_LIN + i) * Kl[(ns + i) * nz + a_] * Kp[(ns + i) * npa + j];
                L->Cz[idx] = acc;
                C0[(long)k * nz * npa + idx] = acc;
            }
            if (lane == 0 && np > 0)
                for (int i = 0; i < nl; i++)
                    for (int p1 = 0; p1 < np; p1++)
                        for (int p2 = 0; p2 < np; p2++)
                            Dp[p1 * npa + p2] += ROW(w, k, S::R_LIN + i) * Kp[(ns + i) * npa + p1] * Kp[(ns + i) * npa + p2];
            sync();
            chol(L->Sz, nz, nz);
            for (int idx = lane; idx < nz * nz; idx += 64) Lz[(long)k * nz * nz + idx] = L->Sz[idx];
            // ---- forward: b-hat columns for the arrow (rhs = C0_k + X_{k-1}' t-hat_{k-1}) ----
            const int m = mnu(k);
            for (int idx = lane; idx < nz * npa; idx += 64) {
                const int a_ = idx / npa, j = idx % npa;
                double acc = L->Cz[idx];
                if (k > 0) {
                    const int mp = mnu(k - 1);
                    for (int r = 0; r < mp; r++) acc += L->X[r * nz + a_] * L->rt[r * NR + 1 + j];
                }
                L->rb[a_ * NR + 1 + j] = acc;
            }
            sync();
            // ---- Y = Lz^-1 Dt' (nz x m) ----
            for (int idx = lane; idx < nz * m; idx += 64) {
                const int j = idx / m, c = idx % m;
                L->Y[j * MNU + c] = Dt(k, c, j);
            }
            sync();
            trsm_lower(L->Sz, nz, nz, L->Y, m, MNU);
            if (np > 0) trsm_lower(L->Sz, nz, nz, L->rb + 1, np, NR);
            for (int idx = lane; idx < nz * m; idx += 64) {
                const int j = idx / m, c = idx % m;
                Yg[(long)k * nz * MNU + j * MNU + c] = L->Y[j * MNU + c];
            }
            // ---- Snu = diag(kinv + reg) + Y'Y ----
            for (int idx = lane; idx < m * m; idx += 64) {
                const int c1 = idx / m, c2 = idx % m;
                double acc = 0.0;
                for (int j = 0; j < nz; j++) acc += L->Y[j * MNU + c1] * L->Y[j * MNU + c2];
                if (c1 == c2) {
                    double ki = 1.0;
                    if (nu_live(k, c1)) {
                        double w1, w2, t1, t2, rxa; bool hg;
                        nu_row_data(k, c1, w, w, W + wo.rx, w1, w2, t1, t2, rxa, hg);
                        ki = hg ? (w1 + w2) / (w1 * w2) : (w1 + w2) / (4.0 * w1 * w2);
                    }
                    kinvg[(long)k * MNU + c1] = ki;
                    acc += ki + (nu_live(k, c1) ? a.reg : 0.0);
                }
                L->Snu[c1 * MNU + c2] = acc;
            }
            sync();
            chol(L->Snu, m, MNU);
            for (int idx = lane; idx < m * m; idx += 64) {
                const int c1 = idx / m, c2 = idx % m;
                Lnu[(long)k * MNU * MNU + c1 * MNU + c2] = L->Snu[c1 * MNU + c2];
            }
            // ---- t-hat columns for the arrow: Lnu^-1 (Ft - Y' b-hat) ----
            if (np > 0) {
                for (int idx = lane; idx < m * np; idx += 64) {
                    const int c = idx / np, j = idx % np;
                    double acc = nu_live(k, c) ? Ft(k, c, j) : 0.0;
                    for (int q = 0; q < nz; q++) acc -= L->Y[q * MNU + c] * L->rb[q * NR + 1 + j];
                    L->rt[c * NR + 1 + j] = acc;
                }
                sync();
                trsm_lower(L->Snu, m, MNU, L->rt + 1, np, NR);
                for (int idx = lane; idx < nz * np; idx += 64) Ycz[(long)k * nz * npa + idx] = L->rb[(idx / np) * NR + 1 + idx % np];
                for (int idx = lane; idx < m * np; idx += 64) Ycnu[(long)k * MNU * npa + idx] = L->rt[(idx / np) * NR + 1 + idx % np];
            }
            // ---- X = Lnu^-1 Et (m x nz): only the dyn rows of Et are non-zero ----
            for (int idx = lane; idx < m * nz; idx += 64) {
                const int c = idx / nz, j = idx % nz;
                L->X[c * nz + j] = (c < nx && k < N - 1) ? Ek(k)[c * nz + j] : 0.0;
            }
            sync();
            trsm_lower(L->Snu, m, MNU, L->X, nz, nz);
            for (int idx = lane; idx < m * nz; idx += 64) Xg[(long)k * MNU * nz + idx] = L->X[idx];
            sync();
        }
        // ---- backward sweep for the arrow columns: (Ycz, Ycnu) <- M^-1 [C0; Ft] ----
        if (np > 0) {
            solve_backward(Ycz, Ycnu, np);
            // Sp = Dp0 - sum_k (C0_k' Ycz_k + Ft_k' Ycnu_k)
            double acc[npa * npa];
            for (int i = 0; i < npa * npa; i++) acc[i] = 0.0;
            for (int k = 0; k < N; k++) {
                for (int r = lane; r < nz + MNU; r += 64) {
                    for (int p1 = 0; p1 < np; p1++) {
                        double coef;
                        const double* yrow;
                        if (r < nz) { coef = C0[(long)k * nz * npa + r * npa + p1]; yrow = Ycz + (long)k * nz * npa + r * npa; }
                        else {
                            const int c = r - nz;
                            if (c >= mnu(k) || !nu_live(k, c)) continue;
                            coef = Ft(k, c, p1); yrow = Ycnu + (long)k * MNU * npa + c * npa;
                        }
                        for (int p2 = 0; p2 < np; p2++) acc[p1 * npa + p2] += coef * yrow[p2];
                    }
                }
            }
            for (int i = 0; i < np; i++)
                for (int j = 0; j < np; j++) {
                    const double t = wave_sum(acc[i * npa + j]);
                    if (lane == 0) {
                        double v = Dp[i * npa + j] - t;
                        if (i == j) {
                            v += P[o.Qp + i];
                            v += typeB_entry(&GROW(w, S::G_TRP0), &GROW(w, S::G_TRP1), np, i, j);
                        } else {
                            v += typeB_entry(&GROW(w, S::G_TRP0), &GROW(w, S::G_TRP1), np, i, j);
                        }
                        for (int q = 0; q < ng; q++) v += GROW(w, S::G_LIN + q) * P[o.Lp + q * npa + i] * P[o.Lp + q * npa + j];
                        L->tmp[i * npa + j] = v;
                    }
                }
            sync();
            // tiny Cholesky of Sp by lane 0 (np is 1 in the current models)
            if (lane == 0) {
                for (int j = 0; j < np; j++) {
                    double d = L->tmp[j * npa + j];
                    for (int q = 0; q < j; q++) d -= L->tmp[j * npa + q] * L->tmp[j * npa + q];
                    if (!(d > 0.0)) { L->fail = 1; d = 1.0; }
                    L->tmp[j * npa + j] = sqrt(d);
                    for (int i = j + 1; i < np; i++) {
                        double v = L->tmp[i * npa + j];
                        for (int q = 0; q < j; q++) v -= L->tmp[i * npa + q] * L->tmp[j * npa + q];
                        L->tmp[i * npa + j] = v / L->tmp[j * npa + j];
                    }
                }
            }
            sync();
            for (int i = 0; i < npa * npa; i++) spL[i] = L->tmp[i];
            sync();
        }
        prof[2] += tick() - t0_;
    }

    // backward sweep: on entry yz[k] holds b-hat_k and ynu[k] holds t-hat_k for `m_` columns (stride npa);
    // on exit they hold z_k and nu_k.
    __device__ void solve_backward(double* yz, double* ynu, int m_)
    {
        const double* Lz = W + wo.Lz; const double* Lnu = W + wo.Lnu; const double* Xg = W + wo.X; const double* Yg = W + wo.Y;
        for (int k = N - 1; k >= 0; k--) {
            const int m = mnu(k);
            double* tz = yz + (long)k * nz * npa;
            double* tn = ynu + (long)k * MNU * npa;
            // nu = Lnu^-T (X z_{k+1} - t-hat)
            for (int idx = lane; idx < m * m_; idx += 64) {
                const int c = idx / m_, j = idx % m_;
                double acc = -tn[c * npa + j];
                if (k < N - 1) {
                    const double* zn = yz + (long)(k + 1) * nz * npa;
                    for (int q = 0; q < nz; q++) acc += Xg[(long)k * MNU * nz + c * nz + q] * zn[q * npa + j];
                }
                L->rt[c * NR + j] = acc;
            }
            sync();
            for (int idx = lane; idx < m * MNU; idx += 64) L->Snu[idx] = Lnu[(long)k * MNU * MNU + idx];
            sync();
            trsm_lowerT(L->Snu, m, MNU, L->rt, m_, NR);
            for (int idx = lane; idx < m * m_; idx += 64) tn[(idx / m_) * npa + idx % m_] = L->rt[(idx / m_) * NR + idx % m_];
            // z = Lz^-T (b-hat - Y nu)
            for (int idx = lane; idx < nz * m_; idx += 64) {
                const int q = idx / m_, j = idx % m_;
                double acc = tz[q * npa + j];
                for (int c = 0; c < m; c++) acc -= Yg[(long)k * nz * MNU + q * MNU + c] * L->rt[c * NR + j];
                L->rb[q * NR + j] = acc;
            }
            for (int idx = lane; idx < nz * nz; idx += 64) L->Sz[idx] = Lz[(long)k * nz * nz + idx];
            sync();
            trsm_lowerT(L->Sz, nz, nz, L->rb, m_, NR);
            for (int idx = lane; idx < nz * m_; idx += 64) tz[(idx / m_) * npa + idx % m_] = L->rb[(idx / m_) * NR + idx % m_];
            sync();
        }
    }

    // ---------------- Newton solve with the stored factorisation ----------------
    // solves (P + G'W^-2 G) dxi = -rxv - G'W^-2 rtil ; writes dxi (main + aux) and nu (nuv)
    __device__ void newton_solve(double* w, double* rtil, double* rxv, double* dxi)
    {
        const long long t0s_ = tick();
        double* fb = W + wo.fb; double* ft = W + wo.ft;  // single-column rhs, stride npa
        double* socW = W + wo.socW;
        const double* Lz = W + wo.Lz; const double* Lnu = W + wo.Lnu; const double* Xg = W + wo.X; const double* Yg = W + wo.Y;
        const double* kinvg = W + wo.kinv;
        double bp[npa];
        for (int j = 0; j < npa; j++) bp[j] = 0.0;  // partial sums per lane, reduced below
        // ---- right-hand sides b_k, t_k and forward substitution ----
        for (int k = 0; k < N; k++) {
            const double* Kl = Klk(k);
            const double* Kp = Kpk(k);
            // cone rows: tl = W^-1 (W^-1 rtil)
            for (int c = lane; c < nsoc; c += 64) {
                const double* Wi = socW + ((long)k * nsoc + c) * 36 + 16;
                double t1[4], t2[4];
                for (int r = 0; r < 4; r++) { double acc = 0.0; for (int q = 0; q < 4; q++) acc += Wi[r * 4 + q] * ROW(rtil, k, S::R_SOC + 4 * c + q); t1[r] = acc; }
                for (int r = 0; r < 4; r++) { double acc = 0.0; for (int q = 0; q < 4; q++) acc += Wi[r * 4 + q] * t1[q]; t2[r] = acc; }
                for (int r = 0; r < 4; r++) L->tmp[4 * c + r] = t2[r];
            }
            sync();
            if (lane < nz) {
                const int j = lane;
                double acc = -Z(rxv, k, j);
                // trust-region block (type B)
                const int j0 = j < nx ? 0 : nx, n = j < nx ? nx : nu;
                double Wt = 0.0, rth = -AUX(rxv, k, j < nx ? S::A_EX : S::A_EU);
                for (int q = 0; q < n; q++) {
                    const double w1 = ROW(w, k, S::R_TR0 + j0 + q), w2 = ROW(w, k, S::R_TR1 + j0 + q);
                    Wt += w1 + w2;
                    rth += w1 * ROW(rtil, k, S::R_TR0 + j0 + q) + w2 * ROW(rtil, k, S::R_TR1 + j0 + q);
                }
                const double w1 = ROW(w, k, S::R_TR0 + j), w2 = ROW(w, k, S::R_TR1 + j);
                acc += -(w1 * ROW(rtil, k, S::R_TR0 + j) - w2 * ROW(rtil, k, S::R_TR1 + j)) + (w1 - w2) * rth / Wt;
                for (int i = 0; i < nl; i++) acc += Kl[(ns + i) * nz + j] * (-ROW(w, k, S::R_LIN + i) * ROW(rtil, k, S::R_LIN + i));
                for (int r = 0; r < 4 * nsoc; r++) acc += Kl[(ns + nl + r) * nz + j] * L->tmp[r];
                L->rb[j * NR] = acc;
            }
            if (np > 0) {
                for (int r = lane; r < nl + 4 * nsoc; r += 64) {
                    const double tl = r < nl ? -ROW(w, k, S::R_LIN + r) * ROW(rtil, k, S::R_LIN + r) : L->tmp[r - nl];
                    for (int j = 0; j < np; j++) bp[j] += Kp[(ns + r) * npa + j] * tl;
                }
            }
            const int m = mnu(k);
            for (int c = lane; c < m; c += 64) {
                double t = 0.0;
                if (nu_live(k, c)) {
                    double w1, w2, t1, t2, rxa; bool hg;
                    nu_row_data(k, c, w, rtil, rxv, w1, w2, t1, t2, rxa, hg);
                    const Pair pr = hg ? pairC(w1, w2, t1, t2, rxa) : pairA(w1, w2, t1, t2, rxa);
                    t = pr.tau / pr.kap;
                }
                L->rt[c * NR] = t;
            }
            sync();
            // bp_k = b_k + X_{k-1}' t-hat_{k-1}
            if (k > 0 && lane < nz) {
                const int mp = mnu(k - 1);
                double acc = L->rb[lane * NR];
                for (int r = 0; r < mp; r++) acc += Xg[(long)(k - 1) * MNU * nz + r * nz + lane] * ft[(long)(k - 1) * MNU * npa + r * npa];
                L->rb[lane * NR] = acc;
            }
            for (int idx = lane; idx < nz * nz; idx += 64) L->Sz[idx] = Lz[(long)k * nz * nz + idx];
            sync();
            trsm_lower(L->Sz, nz, nz, L->rb, 1, NR);
            // t-hat = Lnu^-1 (t - Y' b-hat)
            for (int c = lane; c < m; c += 64) {
                double acc = L->rt[c * NR];
                for (int q = 0; q < nz; q++) acc -= Yg[(long)k * nz * MNU + q * MNU + c] * L->rb[q * NR];
                L->rt[c * NR] = acc;
            }
            for (int idx = lane; idx < m * MNU; idx += 64) L->Snu[idx] = Lnu[(long)k * MNU * MNU + idx];
            sync();
            trsm_lower(L->Snu, m, MNU, L->rt, 1, NR);
            if (lane < nz) fb[(long)k * nz * npa + lane * npa] = L->rb[lane * NR];
            for (int c = lane; c < m; c += 64) ft[(long)k * MNU * npa + c * npa] = L->rt[c * NR];
            sync();
        }
        prof[3] += tick() - t0s_;
        { const long long tb_ = tick(); solve_backward(fb, ft, 1); prof[4] += tick() - tb_; }  // fb <- y_b (z part), ft <- y_b (nu part)
        const long long t1s_ = tick();
        // ---- arrow: dp = Sp^-1 (bp - [C0; Ft]' y_b) ; z -= Ycz dp ; nu -= Ycnu dp ----
        double dp[npa];
        for (int j = 0; j < npa; j++) dp[j] = 0.0;
        if (np > 0) {
            const double* C0 = W + wo.C0; const double* Ycz = W + wo.Ycz; const double* Ycnu = W + wo.Ycnu;
            for (int k = 0; k < N; k++)
                for (int r = lane; r < nz + MNU; r += 64) {
                    double yv;
                    if (r < nz) yv = fb[(long)k * nz * npa + r * npa];
                    else { const int c = r - nz; if (c >= mnu(k) || !nu_live(k, c)) continue; yv = ft[(long)k * MNU * npa + c * npa]; }
                    for (int j = 0; j < np; j++) bp[j] -= (r < nz ? C0[(long)k * nz * npa + r * npa + j] : Ft(k, r - nz, j)) * yv;
                }
            for (int j = 0; j < np; j++) bp[j] = wave_sum(bp[j]);
            // p-only terms: -rxp + trp type-B rhs + glin
            double Wt = 0.0, rth = -GAUX(rxv, S::GA_EP);
            for (int q = 0; q < np; q++) {
                const double w1 = GROW(w, S::G_TRP0 + q), w2 = GROW(w, S::G_TRP1 + q);
                Wt += w1 + w2; rth += w1 * GROW(rtil, S::G_TRP0 + q) + w2 * GROW(rtil, S::G_TRP1 + q);
            }
            for (int j = 0; j < np; j++) {
                const double w1 = GROW(w, S::G_TRP0 + j), w2 = GROW(w, S::G_TRP1 + j);
                double v = bp[j] - PV(rxv, j);
                v += -(w1 * GROW(rtil, S::G_TRP0 + j) - w2 * GROW(rtil, S::G_TRP1 + j)) + (w1 - w2) * rth / Wt;
                for (int q = 0; q < ng; q++) v += P[o.Lp + q * npa + j] * (-GROW(w, S::G_LIN + q) * GROW(rtil, S::G_LIN + q));
                dp[j] = v;
            }
            // solve with chol(Sp)
            for (int i = 0; i < np; i++) { double v = dp[i]; for (int q = 0; q < i; q++) v -= spL[i * npa + q] * dp[q]; dp[i] = v / spL[i * npa + i]; }
            for (int i = np - 1; i >= 0; i--) { double v = dp[i]; for (int q = i + 1; q < np; q++) v -= spL[q * npa + i] * dp[q]; dp[i] = v / spL[i * npa + i]; }
            for (int k = 0; k < N; k++) {
                if (lane < nz) { double v = fb[(long)k * nz * npa + lane * npa]; for (int j = 0; j < np; j++) v -= Ycz[(long)k * nz * npa + lane * npa + j] * dp[j]; fb[(long)k * nz * npa + lane * npa] = v; }
                for (int c = lane; c < mnu(k); c += 64) { double v = ft[(long)k * MNU * npa + c * npa]; for (int j = 0; j < np; j++) v -= Ycnu[(long)k * MNU * npa + c * npa + j] * dp[j]; ft[(long)k * MNU * npa + c * npa] = v; }
            }
            sync();
        }
        // ---- write dz, dp, nu ; recover aux steps ----
        double* nuv = W + wo.nuv;
        for (int k = 0; k < N; k++) {
            if (lane < nz) Z(dxi, k, lane) = fb[(long)k * nz * npa + lane * npa];
            for (int c = lane; c < MNU; c += 64) nuv[(long)k * MNU + c] = (c < mnu(k)) ? ft[(long)k * MNU * npa + c * npa] : 0.0;
        }
        if (lane < npa) PV(dxi, lane) = dp[lane];
        sync();
        (void)kinvg;
        for (int k = 0; k < N; k++) {
            // y_dyn, v : d = (rth + (w1-w2) a)/Wt  |  (rth + w1 a)/Wt
            for (int c = lane; c < nx + ns; c += 64) {
                double val = 0.0;
                if (nu_live(k, c)) {
                    double w1, w2, t1, t2, rxa; bool hg;
                    nu_row_data(k, c, w, rtil, rxv, w1, w2, t1, t2, rxa, hg);
                    const Pair pr = hg ? pairC(w1, w2, t1, t2, rxa) : pairA(w1, w2, t1, t2, rxa);
                    double av = 0.0;
                    for (int j = 0; j < nz; j++) av += Dt(k, c, j) * Z(dxi, k, j);
                    if (c < nx) for (int j = 0; j < nz; j++) av += Ek(k)[c * nz + j] * Z(dxi, k + 1, j);
                    for (int j = 0; j < np; j++) av += Ft(k, c, j) * dp[j];
                    val = (pr.rth + (hg ? w1 : (w1 - w2)) * av) / pr.Wt;
                }
                AUX(dxi, k, c) = val;  // A_Y = 0, A_V = nx: same order as the nu rows
            }
            if (lane < 2) {
                const int j0 = lane == 0 ? 0 : nx, n = lane == 0 ? nx : nu;
                double Wt = 0.0, rth = -AUX(rxv, k, lane == 0 ? S::A_EX : S::A_EU), ha = 0.0;
                for (int q = 0; q < n; q++) {
                    const double w1 = ROW(w, k, S::R_TR0 + j0 + q), w2 = ROW(w, k, S::R_TR1 + j0 + q);
                    Wt += w1 + w2;
                    rth += w1 * ROW(rtil, k, S::R_TR0 + j0 + q) + w2 * ROW(rtil, k, S::R_TR1 + j0 + q);
                    ha += (w1 - w2) * Z(dxi, k, j0 + q);
                }
                AUX(dxi, k, lane == 0 ? S::A_EX : S::A_EU) = (rth + ha) / Wt;
            }
        }
        for (int i = lane; i < AG; i += 64) {
            double val = 0.0;
            if (i < nic + ntc) {
                const bool isic = i < nic;
                const int ii = isic ? i : i - nic, k = isic ? 0 : N - 1, c = nx + ns + ii;
                double w1, w2, t1, t2, rxa; bool hg;
                nu_row_data(k, c, w, rtil, rxv, w1, w2, t1, t2, rxa, hg);
                const Pair pr = pairA(w1, w2, t1, t2, rxa);
                double av = 0.0;
                for (int j = 0; j < nx; j++) av += Dt(k, c, j) * Z(dxi, k, j);
                for (int j = 0; j < np; j++) av += Ft(k, c, j) * dp[j];
                val = (pr.rth + (w1 - w2) * av) / pr.Wt;
            } else if (np > 0) {
                double Wt = 0.0, rth = -GAUX(rxv, S::GA_EP), ha = 0.0;
                for (int q = 0; q < np; q++) {
                    const double w1 = GROW(w, S::G_TRP0 + q), w2 = GROW(w, S::G_TRP1 + q);
                    Wt += w1 + w2; rth += w1 * GROW(rtil, S::G_TRP0 + q) + w2 * GROW(rtil, S::G_TRP1 + q);
                    ha += (w1 - w2) * dp[q];
                }
                val = (rth + ha) / Wt;
            }
            GAUX(dxi, i) = val;
        }
        sync();
        prof[5] += tick() - t1s_;
    }

    // dl = W^-2 (gd + rtil), except penalised pair rows which use nu and the aux dual rows
    __device__ void dlam_from(double* w, double* gd, double* rtil, double* rxv, double* dl) const
    {
        const double* nuv = W + wo.nuv;
        double* socW = W + wo.socW;
        for (int k = 0; k < N; k++) {
            for (int r = lane; r < RS; r += 64) {
                double val;
                if (r < 2 * nx) {
                    const int i = r % nx;
                    if (k < N - 1) { const double nv = nuv[(long)k * MNU + i], rxa = AUX(rxv, k, S::A_Y + i); val = r < nx ? 0.5 * (rxa + nv) : 0.5 * (rxa - nv); }
                    else val = 0.0;
                } else if (r < S::R_TR0) {
                    const int i = (r - S::R_H0) % (ns > 0 ? ns : 1);
                    const double nv = nuv[(long)k * MNU + nx + i], rxa = AUX(rxv, k, S::A_V + i);
                    val = r < S::R_H1 ? nv : rxa - nv;
                } else if (r < S::R_SOC) {
                    val = ROW(w, k, r) * (ROW(gd, k, r) + ROW(rtil, k, r));
                } else {
                    const int c = (r - S::R_SOC) / 4, rr = (r - S::R_SOC) % 4;
                    const double* Wi = socW + ((long)k * nsoc + c) * 36 + 16;
                    double t1[4];
                    for (int q = 0; q < 4; q++) { double acc = 0.0; for (int q2 = 0; q2 < 4; q2++) acc += Wi[q * 4 + q2] * (ROW(gd, k, S::R_SOC + 4 * c + q2) + ROW(rtil, k, S::R_SOC + 4 * c + q2)); t1[q] = acc; }
                    double acc = 0.0;
                    for (int q = 0; q < 4; q++) acc += Wi[rr * 4 + q] * t1[q];
                    val = acc;
                }
                ROW(dl, k, r) = val;
            }
        }
        for (int r = lane; r < RG; r += 64) {
            double val;
            if (r < S::G_TC0) { const int i = r % nic; const double nv = nuv[nx + ns + i], rxa = GAUX(rxv, S::GA_YIC + i); val = r < S::G_IC1 ? 0.5 * (rxa + nv) : 0.5 * (rxa - nv); }
            else if (r < S::G_TRP0) { const int i = (r - S::G_TC0) % ntc; const double nv = nuv[(long)(N - 1) * MNU + nx + ns + i], rxa = GAUX(rxv, S::GA_YTC + i); val = r < S::G_TC1 ? 0.5 * (rxa + nv) : 0.5 * (rxa - nv); }
            else val = GROW(w, r) * (GROW(gd, r) + GROW(rtil, r));
            GROW(dl, r) = val;
        }
        sync();
    }

    // ---------------- NT scaling of the cones ----------------
    __device__ void nt_update(double* s, double* lam)
    {
        double* socW = W + wo.socW;
        for (int idx = lane; idx < N * nsoc; idx += 64) {
            const int k = idx / nsoc, c = idx % nsoc;
            double sv[4], zv[4];
            for (int q = 0; q < 4; q++) { sv[q] = ROW(s, k, S::R_SOC + 4 * c + q); zv[q] = ROW(lam, k, S::R_SOC + 4 * c + q); }
            const double sres = sqrt(sv[0] * sv[0] - sv[1] * sv[1] - sv[2] * sv[2] - sv[3] * sv[3]);
            const double zres = sqrt(zv[0] * zv[0] - zv[1] * zv[1] - zv[2] * zv[2] - zv[3] * zv[3]);
            double sb[4], zb[4], dot = 0.0;
            for (int q = 0; q < 4; q++) { sb[q] = sv[q] / sres; zb[q] = zv[q] / zres; dot += sb[q] * zb[q]; }
            const double gamma = sqrt((1.0 + dot) / 2.0);
            double wb[4];
            wb[0] = (sb[0] + zb[0]) / (2 * gamma);
            for (int q = 1; q < 4; q++) wb[q] = (sb[q] - zb[q]) / (2 * gamma);
            const double eta = sqrt(sres / zres);
            double* Wm = socW + (long)idx * 36;
            double* Wi = Wm + 16;
            double* lt = Wm + 32;
            for (int r = 0; r < 4; r++)
                for (int q = 0; q < 4; q++) {
                    double v;
                    if (r == 0 && q == 0) v = wb[0];
                    else if (r == 0) v = wb[q];
                    else if (q == 0) v = wb[r];
                    else v = (r == q ? 1.0 : 0.0) + wb[r] * wb[q] / (1.0 + wb[0]);
                    Wm[r * 4 + q] = v * eta;
                    Wi[r * 4 + q] = ((r == 0) != (q == 0) ? -v : v) / eta;
                }
            for (int r = 0; r < 4; r++) { double acc = 0.0; for (int q = 0; q < 4; q++) acc += Wm[r * 4 + q] * zv[q]; lt[r] = acc; }
            if (!(sres > 0.0) || !(zres > 0.0) || !isfinite(eta)) L->fail = 1;
        }
        sync();
    }
    __device__ void nt_identity()
    {
        double* socW = W + wo.socW;
        for (int idx = lane; idx < N * nsoc; idx += 64) {
            double* Wm = socW + (long)idx * 36;
            for (int q = 0; q < 16; q++) { Wm[q] = (q % 5 == 0) ? 1.0 : 0.0; Wm[16 + q] = Wm[q]; }
            for (int q = 0; q < 4; q++) Wm[32 + q] = 0.0;
        }
        sync();
    }

    // largest alpha with v + alpha dv in the cone product
    __device__ double max_step(double* v, double* dv) const
    {
        double am = 1e300;
        for (int k = 0; k < N; k++)
            for (int r = lane; r < RS; r += 64) {
                if (!live(k, r)) continue;
                if (r < S::R_SOC) { const double d = ROW(dv, k, r); if (d < 0.0) am = fmin(am, -ROW(v, k, r) / d); }
                else if ((r - S::R_SOC) % 4 == 0) am = fmin(am, soc_step(&ROW(v, k, r), &ROW(dv, k, r)));
            }
        for (int r = lane; r < RG; r += 64) { const double d = GROW(dv, r); if (d < 0.0) am = fmin(am, -GROW(v, r) / d); }
        return wave_min(am);
    }
    __device__ static double soc_step(const double* s, const double* d)
    {
        const double s0 = s[0], d0 = d[0];
        const double dd = d[1] * d[1] + d[2] * d[2] + d[3] * d[3], sd = s[1] * d[1] + s[2] * d[2] + s[3] * d[3],
                     ss = s[1] * s[1] + s[2] * s[2] + s[3] * s[3];
        const double qa = d0 * d0 - dd, qb = 2.0 * (s0 * d0 - sd), qc = s0 * s0 - ss;
        double r1 = -1.0, r2 = -1.0;
        if (fabs(qa) <= 1e-14 * (d0 * d0 + dd + 1e-300)) { if (qb < 0.0) r1 = -qc / qb; }
        else {
            const double disc = qb * qb - 4.0 * qa * qc;
            if (disc >= 0.0) { const double sq = sqrt(disc), qq = -0.5 * (qb + (qb >= 0.0 ? sq : -sq)); r1 = qq / qa; if (qq != 0.0) r2 = qc / qq; }
        }
        double am = 1e300;
        if (r1 > 0.0 && s0 + r1 * d0 >= -1e-12 * (fabs(s0) + fabs(r1 * d0))) am = fmin(am, r1);
        if (r2 > 0.0 && s0 + r2 * d0 >= -1e-12 * (fabs(s0) + fabs(r2 * d0))) am = fmin(am, r2);
        return am;
    }
    // min over rows of the cone margin of v + alpha dv (LP: value; cone: head - ||tail||)
    __device__ double min_margin(double* v, double* dv, double alpha) const
    {
        double mm = 1e300;
        for (int k = 0; k < N; k++)
            for (int r = lane; r < RS; r += 64) {
                if (!live(k, r)) continue;
                if (r < S::R_SOC) mm = fmin(mm, ROW(v, k, r) + (dv ? alpha * ROW(dv, k, r) : 0.0));
                else if ((r - S::R_SOC) % 4 == 0) {
                    double t[4];
                    for (int q = 0; q < 4; q++) t[q] = ROW(v, k, r + q) + (dv ? alpha * ROW(dv, k, r + q) : 0.0);
                    mm = fmin(mm, t[0] - sqrt(t[1] * t[1] + t[2] * t[2] + t[3] * t[3]));
                }
            }
        for (int r = lane; r < RG; r += 64) mm = fmin(mm, GROW(v, r) + (dv ? alpha * GROW(dv, r) : 0.0));
        return wave_min(mm);
    }

    __device__ void run();
};

}  // namespace scp

#include "ipm_run.hpp"
