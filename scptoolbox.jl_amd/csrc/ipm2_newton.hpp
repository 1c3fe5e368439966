// Newton system of the structured IPM: factorisation sweep, solve sweeps, direction recovery.
// Included by ipm2_kernel.hpp.  Algebra: oracle/ipm_struct.py (qd_factor / qd_solve / newton).
//
// The sweeps over the horizon are dependency chains for the single wave that owns a problem, so the
// per-node work is organised to minimise latency rather than flops:
//   * the small Cholesky factorisations and triangular inverses run ENTIRELY IN REGISTERS (one matrix
//     row / column per lane, pivots and multipliers broadcast with DPP row_newbcast) -- no LDS round trips and no
//     barriers inside the O(n) column loop, reciprocal square roots instead of sqrt + divide;
//   * the four mat-vecs of a solve step keep their matrix rows/columns in registers and broadcast the
//     vector with DPP row_newbcast, so a node costs four short FMA chains and ONE barrier (for the staging);
//   * everything that does not depend on the right-hand side (1/kappa, (w1-w2)/Wt, ...) is computed
//     once per factorisation and stored in the node's factor record.
#pragma once
#include <type_traits>

namespace scp {

// Double-precision broadcast of lane `src` (0..15) to the lanes of the FIRST ROW of 16: DPP row_newbcast on the
// two halves.  All dense blocks of the Newton system have dimension <= 16 (static_assert in Ipm2), so every
// producer and consumer lane of a broadcast lives in row 0; lanes 16..63 receive their own row's lane `src`
// (unused).  Compared with v_readlane (VALU -> SGPR -> VALU, two hazard-padded hops per half) this is a plain
// VALU move with no scalar round trip.  `src` must be a compile-time constant after unrolling.
template <int Q>
__device__ __forceinline__ double bc16t(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + Q, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + Q, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rl(double v, int src)
{
    switch (src) {
        case 0: return bc16t<0>(v); case 1: return bc16t<1>(v); case 2: return bc16t<2>(v); case 3: return bc16t<3>(v);
        case 4: return bc16t<4>(v); case 5: return bc16t<5>(v); case 6: return bc16t<6>(v); case 7: return bc16t<7>(v);
        case 8: return bc16t<8>(v); case 9: return bc16t<9>(v); case 10: return bc16t<10>(v); case 11: return bc16t<11>(v);
        case 12: return bc16t<12>(v); case 13: return bc16t<13>(v); case 14: return bc16t<14>(v); default: return bc16t<15>(v);
    }
}
// 1/sqrt(a) and 1/a to full double accuracy from the hardware estimates + Newton steps
__device__ __forceinline__ double fast_rsqrt(double a)
{
    double y = __builtin_amdgcn_rsq(a);
    y = y * (1.5 - 0.5 * a * y * y);
    y = y * (1.5 - 0.5 * a * y * y);
    return y;
}
__device__ __forceinline__ double fast_rcp(double a)
{
    double y = __builtin_amdgcn_rcp(a);
    y = y * (2.0 - a * y);
    y = y * (2.0 - a * y);
    return y;
}

// In-register Cholesky of an n x n SPD matrix stored row-major in LDS (ld).
// On exit Lout (LDS) holds the factor L packed by rows -- entry (i, j < i) at i(i+1)/2 + j -- with the RECIPROCAL of the pivot in
// the diagonal slot (the substitutions multiply).  Returns false on a non-positive pivot.
// Round 4: the factor itself, not its inverse.  Rounds 1-3 kept explicit inverses so that every triangular solve of the sweeps
// was a mat-vec (shorter dependency chains), but the error of L^-1 b through an explicit inverse grows with the CONDITION of L,
// not with the backward-stable substitution's O(eps): in the end-game of a solve (weights lam / s spanning 1e-14 ... 1e14) the
// directions lost the accuracy the last decades of the gap need -- the device took 49 iterations where the scalar CPU
// restatement (substitutions) took 40, with a tail to 150.
template <int n, int ld>
__device__ __forceinline__ bool chol_reg(const double* A, double* Lout, int lane)
{
    double a[n];   // row `lane` of A -> row of L
    double d[n];   // 1 / L_jj (uniform)
    const int ln_ = lane < n ? lane : n - 1;   // unconditional (clamped) LDS reads: lanes >= n carry a copy of the last row
#pragma unroll
    for (int c = 0; c < n; c++) a[c] = A[ln_ * ld + c];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < n; j++) {
        const double ajj = rl(a[j], j);
        ok = ok && (ajj > 0.0 || lane >= 16);   // rows 1..3 of the wave carry no matrix
        const double dj = fast_rsqrt(ajj > 0.0 ? ajj : 1.0);
        d[j] = dj;
        a[j] = a[j] * dj;   // column j of L (lane j: sqrt(ajj))
#pragma unroll
        for (int c = j + 1; c < n; c++) {
            const double lcj = rl(a[j], c);   // L[c][j]
            a[c] -= a[j] * lcj;               // row update (entries with c > lane are never used)
        }
    }
    if (lane < n) {
#pragma unroll
        for (int j = 0; j < n; j++) { if (j <= lane) Lout[lane * (lane + 1) / 2 + j] = (j < lane) ? a[j] : d[j]; }
    }
    return ok;
}
// Forward substitution L y = b across the lanes of DPP row 0: lane i holds b_i on entry, y_i on return; lrow[q] = L[i][q] for
// q < i (0 otherwise), dinv = 1 / L[i][i].
template <int n>
__device__ __forceinline__ double fsub16(double b, const double (&lrow)[n], double dinv)
{
    double acc = b;
#pragma unroll
    for (int q = 0; q < n; q++) {
        const double yq = rl(acc * dinv, q);   // lane q's accumulator is complete: y_q
        acc -= lrow[q] * yq;                   // no-op for the lanes i <= q (lrow[q] = 0)
    }
    return acc * dinv;
}
// Backward substitution L' z = v: lane i holds v_i on entry, z_i on return; lcol[q] = L[q][i] for q > i (0 otherwise).
template <int n>
__device__ __forceinline__ double bsub16(double v, const double (&lcol)[n], double dinv)
{
    double acc = v;
#pragma unroll
    for (int q = n - 1; q >= 0; q--) {
        const double zq = rl(acc * dinv, q);
        acc -= lcol[q] * zq;
    }
    return acc * dinv;
}

// ------------------------------------------------------------------------------------------------
// factor: forward sweep over the nodes.
//   Sz_k  = H0_k + X_{k-1}' X_{k-1}      Lz = chol(Sz)          ("Li" in the record: the factor, reciprocal pivots on the diagonal)
//   Y_k   = Lz^-1 Dt_k'                  Snu_k = diag(1/kappa + reg) + Y'Y,  Ln = chol(Snu)   ("Lni")
//   X_k   = Ln^-1 Et_k                   (substitutions)
// plus the forward-substituted arrow columns (C0_k / Ft_k) and the np x np Schur complement.
// ------------------------------------------------------------------------------------------------
template <class M>
template <int MM>
__device__ __forceinline__ void Ipm2<M>::factor_stage(int k, double* Dp)
{
    constexpr bool MID = (MM == MMID) && (MMID < MNU);   // instantiated for interior nodes only
    double* gC0 = W + wo.C0; double* gYcz = W + wo.Ycz; double* gYcnu = W + wo.Ycnu;
    FPROF_BEGIN();
    // ---- cone rows scaled by W^-1 ----
    for (int idx = lane; idx < 4 * nsoc * nz; idx += 64) {
        const int r = idx / nz, j = idx % nz, c = r / 4, rr = r % 4;
        const double* Wi = L->soc + c * 36 + 16;
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += Wi[rr * 4 + q] * Kl()[(ns + nl + 4 * c + q) * nz + j];
        L->Ysoc[r * nz + j] = acc;
    }
    sync();
    // ---- Sz = H0_k + X'X  (X of the previous node is still in the factor record) ----
#ifdef SCP_K3_MFMA
    // Matrix-core variant (-DSCP_K3_MFMA: the DEFAULT build, K3FLAGS in the Makefile; `make vector` builds the FMA path): Sz = diag + type-B blocks + Z' diag(omega) Z with the rows of Z = [linear
    // rows | W^-1-scaled cone rows | X_{k-1}] as the K dimension of v_mfma_f64_16x16x4_f64 (A[i = lane & 15][k = lane >> 4] =
    // omega_r Z[r][i], B[k][j = lane & 15] = Z[r][j]; D row = (lane >> 4) + 4 reg, column = lane & 15).  One LDS read per lane
    // and K-block instead of two per multiply-add; on gfx950 the f64 matrix rate EQUALS the vector FMA rate (78.6 TFLOP/s), so
    // the only thing to win is operand traffic, and 45 % of the 16 x 16 x 24 tile is padding.  Measured: DESIGN.md section 4.1.
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc4 = {0.0, 0.0, 0.0, 0.0};
        const int col = lane & 15, kq = lane >> 4, cc = col < nz ? col : nz - 1;
        const int mp = k == 0 ? 0 : (k == 1 ? MNU : MMID);
        const int nzr = nl + 4 * nsoc + mp;
#pragma unroll 1
        for (int r0_ = 0; r0_ < nzr; r0_ += 4) {
            const int r = r0_ + kq;
            double zb = 0.0, om = 1.0;
            if (r < nl) { zb = Kl()[(ns + r) * nz + cc]; om = L->r0[S::R_LIN + r]; }
            else if (r < nl + 4 * nsoc) zb = L->Ysoc[(r - nl) * nz + cc];
            else if (r < nzr) zb = (mp == MNU ? Xm(MNU) : Xm(MMID))[(r - nl - 4 * nsoc) * nz + cc];
            if (col >= nz) zb = 0.0;
            acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(om * zb, zb, acc4, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int a_ = kq + 4 * q, b_ = col;
            if (a_ < nz && b_ < nz) {
                double acc = acc4[q] + ((a_ == b_) ? L->Pk[S::O_QD + a_] : 0.0);
                const bool ax = a_ < nx, bx = b_ < nx;
                if (ax && bx) acc += typeB_entry<nx>(L->r0 + S::R_TR0, L->r0 + S::R_TR1, a_, b_);
                else if (!ax && !bx) acc += typeB_entry<nu>(L->r0 + S::R_TR0 + nx, L->r0 + S::R_TR1 + nx, a_ - nx, b_ - nx);
                L->Sz[a_ * nz + b_] = acc;
            }
        }
    }
#else
    for (int idx = lane; idx < nz * nz; idx += 64) {
        const int a_ = idx / nz, b_ = idx % nz;
        double acc = (a_ == b_) ? L->Pk[S::O_QD + a_] : 0.0;
        const bool ax = a_ < nx, bx = b_ < nx;
        if (ax && bx) acc += typeB_entry<nx>(L->r0 + S::R_TR0, L->r0 + S::R_TR1, a_, b_);
        else if (!ax && !bx) acc += typeB_entry<nu>(L->r0 + S::R_TR0 + nx, L->r0 + S::R_TR1 + nx, a_ - nx, b_ - nx);
#pragma unroll
        for (int i = 0; i < nl; i++) acc += L->r0[S::R_LIN + i] * Kl()[(ns + i) * nz + a_] * Kl()[(ns + i) * nz + b_];
#pragma unroll
        for (int r = 0; r < 4 * nsoc; r++) acc += L->Ysoc[r * nz + a_] * L->Ysoc[r * nz + b_];
        if (k > 0) {
            if (k == 1) {
#pragma unroll
                for (int r = 0; r < MNU; r++) acc += Xm(MNU)[r * nz + a_] * Xm(MNU)[r * nz + b_];
            } else {
#pragma unroll
                for (int r = 0; r < MMID; r++) acc += Xm(MMID)[r * nz + a_] * Xm(MMID)[r * nz + b_];
            }
        }
        L->Sz[idx] = acc;
    }
#endif
    // ---- C0_k and forward substitution of the arrow columns ----
    if (np > 0) {
        for (int idx = lane; idx < nz * np; idx += 64) {
            const int a_ = idx / np, j = idx % np;
            double c0 = 0.0;
#pragma unroll
            for (int i = 0; i < nl; i++) c0 += L->r0[S::R_LIN + i] * Kl()[(ns + i) * nz + a_] * Kp()[(ns + i) * npa + j];
            gC0[(long)k * nz * npa + a_ * npa + j] = c0;
            double acc = c0;
            if (k > 0) {
                const int mp = mnu(k - 1);
                for (int r = 0; r < mp; r++) acc += Xm(mp)[r * nz + a_] * L->ct[r * npa + j];
            }
            L->Cz[a_ * npa + j] = acc;
        }
        if (lane == 0)
            for (int i = 0; i < nl; i++)
                for (int p1 = 0; p1 < np; p1++)
                    for (int p2 = 0; p2 < np; p2++)
                        Dp[p1 * npa + p2] += L->r0[S::R_LIN + i] * Kp()[(ns + i) * npa + p1] * Kp()[(ns + i) * npa + p2];
    }
    sync();
    FPROF(0);
    // ---- Li = chol(Sz)^-1 in registers ----
    if (!chol_reg<nz, nz>(L->Sz, Li(), lane)) L->fail = 1;
    sync();
    FPROF(1);
    // ---- Y = Lz^-1 Dt' : lane c owns column c (c < MM) ; lanes MM..MM+np-1 do the arrow columns cb = Lz^-1 Cz ----
    // (forward substitution inside the lane; the entries of L are uniform LDS reads)
    {
        double dt[nz], y[nz];
        const bool isY = lane < MM, isC = (np > 0) && lane >= MM && lane < MM + np;
#pragma unroll
        for (int q = 0; q < nz; q++) dt[q] = isY ? Dt_m<MID>(k, lane, q) : (isC ? L->Cz[q * npa + (lane - MM)] : 0.0);
#pragma unroll
        for (int j = 0; j < nz; j++) {
            double acc = dt[j];
#pragma unroll
            for (int q = 0; q < j; q++) acc -= Li()[j * (j + 1) / 2 + q] * y[q];
            y[j] = acc * Li()[j * (j + 1) / 2 + j];
        }
        if (isY) {
#pragma unroll
            for (int j = 0; j < nz; j++) Ym(MM)[j * MM + lane] = y[j];
        } else if (isC) {
#pragma unroll
            for (int j = 0; j < nz; j++) { L->cb[j * npa + (lane - MM)] = y[j]; gYcz[(long)k * nz * npa + j * npa + (lane - MM)] = y[j]; }
        }
    }
    sync();
    FPROF(3);
    // ---- per-row elimination coefficients (independent of the right-hand side) + Snu ----
    for (int c = lane; c < MM; c += 64) {
        double w1 = 1.0, w2 = 1.0, t1, t2, rxa; bool hg = false;
        const bool lv = nu_live_m<MID>(k, c);
        if (lv) nu_row_data(k, c, L->r0, L->r0, L->g0, L->g0, L->ak, L->ga, w1, w2, t1, t2, rxa, hg);
        const double iWt = fast_rcp(w1 + w2);
        const double kap = (hg ? 1.0 : 4.0) * w1 * w2 * iWt;
        double* cf = Cf(MM) + c * 2;
        cf[0] = lv ? (hg ? w1 : (w1 - w2)) * iWt : 0.0;   // coefficient of rth in tau
        cf[1] = lv ? fast_rcp(kap) : 1.0;                 // 1/kappa (1 for absent rows: identity pivot)
    }
    // arrow right-hand side Ft - Y' cb, one (row, column) per lane (consumed by the X stage below)
    if (np > 0) {
        for (int idx = lane; idx < MM * np; idx += 64) {
            const int q = idx / np, j = idx % np;
            double v = nu_live_m<MID>(k, q) ? Ft_m<MID>(k, q, j) : 0.0;
#pragma unroll
            for (int i = 0; i < nz; i++) v -= Ym(MM)[i * MM + q] * L->cb[i * npa + j];
            L->tmp[q * npa + j] = v;
        }
    }
    sync();
#ifdef SCP_K3_MFMA
    {   // Snu = diag(1/kappa + reg) + Y'Y on the matrix core: K = the nz rows of Y
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc4 = {0.0, 0.0, 0.0, 0.0};
        const int col = lane & 15, kq = lane >> 4, cc = col < MM ? col : MM - 1;
#pragma unroll
        for (int j0 = 0; j0 < nz; j0 += 4) {
            const int j = j0 + kq;
            double yb = Ym(MM)[(j < nz ? j : nz - 1) * MM + cc];
            if (j >= nz || col >= MM) yb = 0.0;
            acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(yb, yb, acc4, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c1 = kq + 4 * q, c2 = col;
            if (c1 < MM && c2 < MM) {
                double acc = acc4[q];
                if (c1 == c2) acc += Cf(MM)[c1 * 2 + 1] + (nu_live_m<MID>(k, c1) ? a.reg : 0.0);
                L->Snu[c1 * MNU + c2] = acc;
            }
        }
    }
#else
    for (int idx = lane; idx < MM * MM; idx += 64) {
        const int c1 = idx / MM, c2 = idx % MM;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < nz; j++) acc += Ym(MM)[j * MM + c1] * Ym(MM)[j * MM + c2];
        if (c1 == c2) acc += Cf(MM)[c1 * 2 + 1] + (nu_live_m<MID>(k, c1) ? a.reg : 0.0);
        L->Snu[c1 * MNU + c2] = acc;
    }
#endif
    sync();
    FPROF(4);
    if (!chol_reg<MM, MNU>(L->Snu, Lni(MM), lane)) L->fail = 1;
    sync();
    FPROF(5);
    // ---- X = Ln^-1 Et : lane j owns column j (j < nz) ; arrow: ct = Ln^-1 (Ft - Y' cb) on lanes nz..nz+np-1 ----
    {
        const bool isX = lane < nz, isC = (np > 0) && lane >= nz && lane < nz + np;
        double e[MM];
#pragma unroll
        for (int q = 0; q < MM; q++) {
            double v = 0.0;
            if (isX) v = (q < nx) ? E()[q * nz + lane] : 0.0;
            else if (isC) v = L->tmp[q * npa + (lane - nz)];
            e[q] = v;
        }
        double x[MM];
#pragma unroll
        for (int c = 0; c < MM; c++) {
            double acc = e[c];
#pragma unroll
            for (int q = 0; q < c; q++) acc -= Lni(MM)[c * (c + 1) / 2 + q] * x[q];
            x[c] = acc * Lni(MM)[c * (c + 1) / 2 + c];
        }
        sync();   // all reads of the previous X / ct are done before they are overwritten
        if (isX) {
#pragma unroll
            for (int c = 0; c < MM; c++) Xm(MM)[c * nz + lane] = x[c];
        } else if (isC) {
#pragma unroll
            for (int c = 0; c < MM; c++) { L->ct[c * npa + (lane - nz)] = x[c]; gYcnu[(long)k * MNU * npa + c * npa + (lane - nz)] = x[c]; }
        }
    }
    sync();
    FPROF(6);
    storeF(k);
}

template <class M>
__device__ __forceinline__ void Ipm2<M>::factor(double* w)
{
    const long long t0_ = tick();
    const double* socW = W + wo.socW;
    double* gC0 = W + wo.C0; double* gYcz = W + wo.Ycz; double* gYcnu = W + wo.Ycnu;
    double Dp[npa * npa];
#pragma unroll
    for (int i = 0; i < npa * npa; i++) Dp[i] = 0.0;  // lane 0 accumulates
    load_grows(L->g0, w);
    (void)socW;
    prefetch(0); pf_rows(pR0, w, 0); pf_soc(0);
    // the two boundary nodes are peeled off so that the hot loop holds a single (mid-node) instantiation
    auto node_head = [&](int k) {
        commit(); cm_rows(L->r0, pR0); cm_soc();
        sync();
        if (k + 1 < N) { prefetch(k + 1); pf_rows(pR0, w, k + 1); pf_soc(k + 1); }
    };
    node_head(0); factor_stage<MNU>(0, Dp); sync();
#pragma unroll 1
    for (int k = 1; k < N - 1; k++) { node_head(k); factor_stage<MMID>(k, Dp); sync(); }
    if (N > 1) { node_head(N - 1); factor_stage<MNU>(N - 1, Dp); sync(); }
    gsync();
    // ---- arrow: back-substitute the np columns, then Sp = Dp0 - [C0; Ft]' Yc, chol(Sp) ----
    if (np > 0) {
        solve_backward_cols();
        double acc[npa * npa];
#pragma unroll
        for (int i = 0; i < npa * npa; i++) acc[i] = 0.0;
        prefetch_ft(0);
        for (int k = 0; k < N; k++) {
            commit_ft();
            sync();
            if (k + 1 < N) prefetch_ft(k + 1);
            for (int r = lane; r < nz + MNU; r += 64) {
#pragma unroll
                for (int p1 = 0; p1 < np; p1++) {
                    double coef;
                    const double* yrow;
                    if (r < nz) { coef = gC0[(long)k * nz * npa + r * npa + p1]; yrow = gYcz + (long)k * nz * npa + r * npa; }
                    else {
                        const int c = r - nz;
                        if (c >= mnu(k) || !nu_live(k, c)) continue;
                        coef = Ft(k, c, p1); yrow = gYcnu + (long)k * MNU * npa + c * npa;
                    }
#pragma unroll
                    for (int p2 = 0; p2 < np; p2++) acc[p1 * npa + p2] += coef * yrow[p2];
                }
            }
            sync();
        }
        for (int i = 0; i < np; i++)
            for (int j = 0; j < np; j++) {
                const double t = wave_sum(acc[i * npa + j]);
                if (lane == 0) {
                    double v = Dp[i * npa + j] - t;
                    if (i == j) v += L->G[S::Q_QP + i];
                    v += typeB_entry<(np > 0 ? np : 1)>(L->g0 + S::G_TRP0, L->g0 + S::G_TRP1, i, j);
                    for (int q = 0; q < ng; q++) v += L->g0[S::G_LIN + q] * gLp()[q * npa + i] * gLp()[q * npa + j];
                    L->tmp[i * npa + j] = v;
                }
            }
        sync();
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
        for (int i = lane; i < npa * npa; i += 64) L->spL[i] = L->tmp[i];
        gsync();
    }
    PROF_ADD2(2, tick() - t0_);
#ifdef SCP_FACTOR_PROF
    if (lane == 0) for (int i = 0; i < 8; i++) if (i != 2 && i != 7) L->prof[i] += fprof_[i];
#endif
}

// backward sweep for the np arrow columns held in (Ycz = b-hat, Ycnu = t-hat)
template <class M>
__device__ __forceinline__ void Ipm2<M>::solve_backward_cols()
{
    double* yz = W + wo.Ycz; double* yn = W + wo.Ycnu;
    if constexpr (np == 1) {
        // one arrow column: exactly a backward solve with (b-hat, t-hat) = (Ycz, Ycnu), done in place with the
        // register mat-vec chain of the Newton solves (boundary nodes peeled)
        double zn = 0.0, bh_ = 0.0, th_ = 0.0;
        prefetchF(N - 1);
        pB1 = yz[(long)(N - 1) * nz + (lane < nz ? lane : nz - 1)];
        pB2 = yn[(long)(N - 1) * MNU + (lane < MNU ? lane : MNU - 1)];
        auto head = [&](int k) {
            commitF();
            bh_ = pB1; th_ = pB2;
            sync();
            if (k > 0) {
                prefetchF(k - 1);
                pB1 = yz[(long)(k - 1) * nz + (lane < nz ? lane : nz - 1)];
                pB2 = yn[(long)(k - 1) * MNU + (lane < MNU ? lane : MNU - 1)];
            }
        };
        head(N - 1); zn = bwd_stage<MNU>(N - 1, zn, bh_, th_, yz, yn); sync();
#pragma unroll 1
        for (int k = N - 2; k >= 1; k--) { head(k); zn = bwd_stage<MMID>(k, zn, bh_, th_, yz, yn); sync(); }
        if (N > 1) { head(0); zn = bwd_stage<MNU>(0, zn, bh_, th_, yz, yn); sync(); }
        gsync();
    } else if constexpr (np > 1) {
    prefetchF(N - 1);
    for (int k = N - 1; k >= 0; k--) {
        const int m = mnu(k);
        commitF();
        for (int idx = lane; idx < nz * np; idx += 64) L->cb[(idx / np) * npa + idx % np] = yz[(long)k * nz * npa + (idx / np) * npa + idx % np];
        for (int idx = lane; idx < m * np; idx += 64) L->ct[(idx / np) * npa + idx % np] = yn[(long)k * MNU * npa + (idx / np) * npa + idx % np];
        sync();
        if (k > 0) prefetchF(k - 1);
        // u = X z_{k+1} - t-hat
        for (int idx = lane; idx < m * np; idx += 64) {
            const int c = idx / np, j = idx % np;
            double acc = -L->ct[c * npa + j];
            if (k < N - 1) {
#pragma unroll
                for (int q = 0; q < nz; q++) acc += Xm(m)[c * nz + q] * L->Cz[q * npa + j];   // Cz holds z_{k+1} columns
            }
            L->tmp[c * npa + j] = acc;
        }
        sync();
        // nu = Ln^-T u  (backward substitution, one lane per column)
        for (int j = lane; j < np; j += 64) {
            for (int c = m - 1; c >= 0; c--) {
                double acc = L->tmp[c * npa + j];
                for (int r = c + 1; r < m; r++) acc -= Lni(m)[r * (r + 1) / 2 + c] * L->ct[r * npa + j];
                L->ct[c * npa + j] = acc * Lni(m)[c * (c + 1) / 2 + c];
            }
        }
        sync();
        for (int idx = lane; idx < m * np; idx += 64) yn[(long)k * MNU * npa + (idx / np) * npa + idx % np] = L->ct[(idx / np) * npa + idx % np];
        // v = b-hat - Y nu ; z = Li' v
        for (int idx = lane; idx < nz * np; idx += 64) {
            const int q = idx / np, j = idx % np;
            double acc = L->cb[q * npa + j];
            for (int c = 0; c < m; c++) acc -= Ym(m)[q * m + c] * L->ct[c * npa + j];
            L->tmp[q * npa + j] = acc;
        }
        sync();
        for (int j = lane; j < np; j += 64) {      // z = Lz^-T v
            for (int q = nz - 1; q >= 0; q--) {
                double acc = L->tmp[q * npa + j];
                for (int r = q + 1; r < nz; r++) acc -= Li()[r * (r + 1) / 2 + q] * L->Cz[r * npa + j];
                acc *= Li()[q * (q + 1) / 2 + q];
                L->Cz[q * npa + j] = acc;
                yz[(long)k * nz * npa + q * npa + j] = acc;
            }
        }
        sync();
    }
    gsync();
    }
}

// ------------------------------------------------------------------------------------------------
// newton_solve: (P + G'W^-2 G) dxi = -rxv - G'W^-2 rtil with the stored factorisation.
// Writes the MAIN part of dxi (dz, dp) and nu; finish_direction() completes aux / dlam.
// ------------------------------------------------------------------------------------------------
template <class M>
template <int MM>
__device__ __forceinline__ double Ipm2<M>::fwd_stage(int k, double znx, double* bp)
{
    constexpr bool MID = (MM == MMID) && (MMID < MNU);
    double* fb = W + wo.fb; double* ft = W + wo.ft;
    // ---- matrix rows / columns of this node into registers (LDS reads pipeline) ----
    double li[nz], yc[nz], lni[MM], xc[MM];
    // unconditional LDS reads with clamped lane indices (no exec-mask branches); lanes outside a block compute
    // values nobody reads
    const int lz_ = lane < nz ? lane : nz - 1, lm_ = lane < MM ? lane : MM - 1;
    const int tz_ = lz_ * (lz_ + 1) / 2, tm_ = lm_ * (lm_ + 1) / 2;   // row starts in the packed triangles
#pragma unroll
    for (int q = 0; q < nz; q++) { const double v = Li()[tz_ + (q < lz_ ? q : lz_)]; li[q] = q < lz_ ? v : 0.0; yc[q] = Ym(MM)[q * MM + lm_]; }
#pragma unroll
    for (int q = 0; q < MM; q++) { const double v = Lni(MM)[tm_ + (q < lm_ ? q : lm_)]; lni[q] = q < lm_ ? v : 0.0; xc[q] = Xm(MM)[q * nz + lz_]; }
    const double dz_ = Li()[tz_ + lz_], dn_ = Lni(MM)[tm_ + lm_];   // reciprocal pivots of this lane's rows
    // ---- cone rows: tl = W^-1 (W^-1 rtil)  (lanes 0..nsoc-1), staged through LDS tmp ----
    for (int c = lane; c < nsoc; c += 64) {
        const double* Wi = L->soc + c * 36 + 16;
        double t1[4], t2[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { double acc = 0.0; for (int q = 0; q < 4; q++) acc += Wi[r * 4 + q] * L->r1[S::R_SOC + 4 * c + q]; t1[r] = acc; }
#pragma unroll
        for (int r = 0; r < 4; r++) { double acc = 0.0; for (int q = 0; q < 4; q++) acc += Wi[r * 4 + q] * t1[q]; t2[r] = acc; }
#pragma unroll
        for (int r = 0; r < 4; r++) L->tmp[4 * c + r] = t2[r];
    }
    if (nsoc > 0) sync();
    // ---- right-hand sides: b (lane j < nz), t (lane c < MM) ----
    double b = 0.0, t = 0.0;
    if (lane < nz) {
        const int j = lane;
        double acc = -L->zk[j] + znx;
        const bool isx = j < nx;
        double rth = -L->ak[isx ? S::A_EX : S::A_EU], Wt = 0.0;
        if (isx) {
#pragma unroll
            for (int q = 0; q < nx; q++) { const double w1 = L->r0[S::R_TR0 + q], w2 = L->r0[S::R_TR1 + q]; Wt += w1 + w2; rth += w1 * L->r1[S::R_TR0 + q] + w2 * L->r1[S::R_TR1 + q]; }
        } else {
#pragma unroll
            for (int q = nx; q < nz; q++) { const double w1 = L->r0[S::R_TR0 + q], w2 = L->r0[S::R_TR1 + q]; Wt += w1 + w2; rth += w1 * L->r1[S::R_TR0 + q] + w2 * L->r1[S::R_TR1 + q]; }
        }
        const double w1 = L->r0[S::R_TR0 + j], w2 = L->r0[S::R_TR1 + j];
        acc += -(w1 * L->r1[S::R_TR0 + j] - w2 * L->r1[S::R_TR1 + j]) + (w1 - w2) * rth * fast_rcp(Wt);
#pragma unroll
        for (int i = 0; i < nl; i++) acc += Kl()[(ns + i) * nz + j] * (-L->r0[S::R_LIN + i] * L->r1[S::R_LIN + i]);
#pragma unroll
        for (int r = 0; r < 4 * nsoc; r++) acc += Kl()[(ns + nl + r) * nz + j] * L->tmp[r];
        b = acc;
    }
    if (lane < MM) {
        const int c = lane;
        if (nu_live_m<MID>(k, c)) {
            double w1, w2, t1, t2, rxa; bool hg;
            nu_row_data(k, c, L->r0, L->r1, L->g0, L->g1, L->ak, L->ga, w1, w2, t1, t2, rxa, hg);
            const double* cf = Cf(MM) + c * 2;
            const double r1 = w1 * t1, r2 = w2 * t2;
            const double rth = -rxa + r1 + r2;
            // tau = -(r1 - r2) + (w1-w2) rth / Wt  (type A)   |   -r1 + w1 rth / Wt  (hinge)
            t = (-(r1 - (hg ? 0.0 : r2)) + cf[0] * rth) * cf[1];
        }
    }
    if (np > 0) {
        for (int r = lane; r < nl + 4 * nsoc; r += 64) {
            const double tl = r < nl ? -L->r0[S::R_LIN + r] * L->r1[S::R_LIN + r] : L->tmp[r - nl];
#pragma unroll
            for (int j = 0; j < np; j++) bp[j] += Kp()[(ns + r) * npa + j] * tl;
        }
    }
    // ---- chain: b-hat = Lz^-1 b ; tp = t - Y' b-hat ; t-hat = Ln^-1 tp ; znx' = X' t-hat ----
    const double bh = fsub16<nz>(b, li, dz_);
    double tp = t;
#pragma unroll
    for (int q = 0; q < nz; q++) tp -= yc[q] * rl(bh, q);
    const double th = fsub16<MM>(tp, lni, dn_);
    double zx = 0.0;
#pragma unroll
    for (int r = 0; r < MM; r++) zx += xc[r] * rl(th, r);
    if (lane < nz) fb[(long)k * nz + lane] = bh;
    if (lane < MNU) ft[(long)k * MNU + lane] = (lane < MM) ? th : 0.0;
    return zx;
}

template <class M>
template <int MM>
__device__ __forceinline__ double Ipm2<M>::bwd_stage(int k, double zn, double bh_in, double th_in, double* zo, double* nuo)
{
    double xr[nz], lnc[MM], yr[MM], lic[nz];
    const int lz_ = lane < nz ? lane : nz - 1, lm_ = lane < MM ? lane : MM - 1;
#pragma unroll
    for (int q = 0; q < nz; q++) { xr[q] = Xm(MM)[lm_ * nz + q]; const double v = Li()[q * (q + 1) / 2 + (lz_ < q ? lz_ : q)]; lic[q] = lz_ < q ? v : 0.0; }
#pragma unroll
    for (int q = 0; q < MM; q++) { const double v = Lni(MM)[q * (q + 1) / 2 + (lm_ < q ? lm_ : q)]; lnc[q] = lm_ < q ? v : 0.0; yr[q] = Ym(MM)[lz_ * MM + q]; }
    const double dz_ = Li()[lz_ * (lz_ + 1) / 2 + lz_], dn_ = Lni(MM)[lm_ * (lm_ + 1) / 2 + lm_];   // reciprocal pivots
    const double bh = (lane < nz) ? bh_in : 0.0;
    const double th = (lane < MM) ? th_in : 0.0;
    // u = X z+ - t-hat ; nu = Ln^-T u ; v = b-hat - Y nu ; z = Lz^-T v
    double u = -th;
#pragma unroll
    for (int q = 0; q < nz; q++) u += xr[q] * rl(zn, q);
    const double nu_ = bsub16<MM>(u, lnc, dn_);
    double v = bh;
#pragma unroll
    for (int c = 0; c < MM; c++) v -= yr[c] * rl(nu_, c);
    const double z = bsub16<nz>(v, lic, dz_);
    if (lane < nz) zo[(long)k * nz + lane] = z;
    if (lane < MNU) nuo[(long)k * MNU + lane] = (lane < MM) ? nu_ : 0.0;
    return z;
}

template <class M>
__device__ __forceinline__ void Ipm2<M>::newton_solve(double* w, double* rtil, double* rxv, double* dxi)
{
    const long long t0s_ = tick();
    const double* socW = W + wo.socW;
    double* fb = W + wo.fb; double* ft = W + wo.ft; double* nuv = W + wo.nuv;
    double bp[npa];
#pragma unroll
    for (int j = 0; j < npa; j++) bp[j] = 0.0;
    load_grows(L->g0, w); load_grows(L->g1, rtil);
    for (int i = lane; i < AG; i += 64) L->ga[i] = GAUX(rxv, i);
    double znx = 0.0;   // lane j: (X_{k-1}' t-hat_{k-1})_j
    (void)socW;
    gsync();
    prefetch_r<S::O_KL, SR>(0); prefetchF(0); pf_rows(pR0, w, 0); pf_rows(pR1, rtil, 0); pf_soc(0);   // the forward sweep reads Kl, Kp only
    pZ = Z(rxv, 0, lane < nz ? lane : nz - 1); pA = AUX(rxv, 0, lane < AS ? lane : AS - 1);
    // boundary nodes peeled off: the hot loop holds the mid-node instantiation only
    auto fwd_head = [&](int k) {
        commit_r<S::O_KL, SR>(); commitF(); cm_rows(L->r0, pR0); cm_rows(L->r1, pR1); cm_soc();
        if (lane < nz) L->zk[lane] = pZ;
        if (lane < AS) L->ak[lane] = pA;
        sync();
        if (k + 1 < N) {
            prefetch_r<S::O_KL, SR>(k + 1); prefetchF(k + 1); pf_rows(pR0, w, k + 1); pf_rows(pR1, rtil, k + 1); pf_soc(k + 1);
            pZ = Z(rxv, k + 1, lane < nz ? lane : nz - 1); pA = AUX(rxv, k + 1, lane < AS ? lane : AS - 1);
        }
    };
    fwd_head(0); znx = fwd_stage<MNU>(0, znx, bp); sync();
#pragma unroll 1
    for (int k = 1; k < N - 1; k++) { fwd_head(k); znx = fwd_stage<MMID>(k, znx, bp); sync(); }
    if (N > 1) { fwd_head(N - 1); znx = fwd_stage<MNU>(N - 1, znx, bp); sync(); }
    gsync();
    PROF_ADD(3, tick() - t0s_);
    const long long tb_ = tick();
    // ---------------- backward sweep ----------------
    double zn = 0.0;
    prefetchF(N - 1);
    pB1 = fb[(long)(N - 1) * nz + (lane < nz ? lane : nz - 1)];
    pB2 = ft[(long)(N - 1) * MNU + (lane < MNU ? lane : MNU - 1)];
    double bh_ = 0.0, th_ = 0.0;
    auto bwd_head = [&](int k) {
        commitF();
        bh_ = pB1; th_ = pB2;
        sync();
        if (k > 0) {
            prefetchF(k - 1);
            pB1 = fb[(long)(k - 1) * nz + (lane < nz ? lane : nz - 1)];
            pB2 = ft[(long)(k - 1) * MNU + (lane < MNU ? lane : MNU - 1)];
        }
    };
    bwd_head(N - 1); zn = bwd_stage<MNU>(N - 1, zn, bh_, th_, dxi, nuv); sync();
#pragma unroll 1
    for (int k = N - 2; k >= 1; k--) { bwd_head(k); zn = bwd_stage<MMID>(k, zn, bh_, th_, dxi, nuv); sync(); }
    if (N > 1) { bwd_head(0); zn = bwd_stage<MNU>(0, zn, bh_, th_, dxi, nuv); sync(); }
    gsync();
    PROF_ADD(4, tick() - tb_);
    const long long t1s_ = tick();
    // ---------------- arrow: dp = Sp^-1 (bp - [C0; Ft]' y_b) ; z -= Ycz dp ; nu -= Ycnu dp ----------------
    double dp[npa];
#pragma unroll
    for (int j = 0; j < npa; j++) dp[j] = 0.0;
    if (np > 0) {
        const double* gC0 = W + wo.C0; const double* gYcz = W + wo.Ycz; const double* gYcnu = W + wo.Ycnu;
        prefetch_ft(0);
        for (int k = 0; k < N; k++) {
            commit_ft();
            sync();
            if (k + 1 < N) prefetch_ft(k + 1);
            for (int r = lane; r < nz + MNU; r += 64) {
                double yv;
                if (r < nz) yv = dxi[(long)k * nz + r];
                else { const int c = r - nz; if (c >= mnu(k) || !nu_live(k, c)) continue; yv = nuv[(long)k * MNU + c]; }
#pragma unroll
                for (int j = 0; j < np; j++) bp[j] -= (r < nz ? gC0[(long)k * nz * npa + r * npa + j] : Ft(k, r - nz, j)) * yv;
            }
            sync();
        }
#pragma unroll
        for (int j = 0; j < np; j++) bp[j] = wave_sum(bp[j]);
        double Wt = 0.0, rth = -L->ga[S::GA_EP];
        for (int q = 0; q < np; q++) {
            const double w1 = L->g0[S::G_TRP0 + q], w2 = L->g0[S::G_TRP1 + q];
            Wt += w1 + w2; rth += w1 * L->g1[S::G_TRP0 + q] + w2 * L->g1[S::G_TRP1 + q];
        }
        for (int j = 0; j < np; j++) {
            const double w1 = L->g0[S::G_TRP0 + j], w2 = L->g0[S::G_TRP1 + j];
            double v = bp[j] - PV(rxv, j);
            v += -(w1 * L->g1[S::G_TRP0 + j] - w2 * L->g1[S::G_TRP1 + j]) + (w1 - w2) * rth / Wt;
            for (int q = 0; q < ng; q++) v += gLp()[q * npa + j] * (-L->g0[S::G_LIN + q] * L->g1[S::G_LIN + q]);
            dp[j] = v;
        }
        for (int i = 0; i < np; i++) { double v = dp[i]; for (int q = 0; q < i; q++) v -= L->spL[i * npa + q] * dp[q]; dp[i] = v / L->spL[i * npa + i]; }
        for (int i = np - 1; i >= 0; i--) { double v = dp[i]; for (int q = i + 1; q < np; q++) v -= L->spL[q * npa + i] * dp[q]; dp[i] = v / L->spL[i * npa + i]; }
        // z -= Ycz dp ; nu -= Ycnu dp  (batched: 4 elements per lane in flight)
        for (int part = 0; part < 2; part++) {
            double* v_ = part == 0 ? dxi : nuv;
            const double* y_ = part == 0 ? gYcz : gYcnu;
            const long n = (long)N * (part == 0 ? nz : MNU);
            for (long base = lane; base < n; base += 256) {
                double v[4], y[4][npa];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    long idx = base + 64 * u; idx = idx < n ? idx : n - 1;
                    v[u] = v_[idx];
#pragma unroll
                    for (int j = 0; j < np; j++) y[u][j] = y_[idx * npa + j];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const long idx = base + 64 * u;
                    double acc = v[u];
#pragma unroll
                    for (int j = 0; j < np; j++) acc -= y[u][j] * dp[j];
                    if (idx < n) v_[idx] = acc;
                }
            }
        }
    }
    if (lane < npa) PV(dxi, lane) = dp[lane];
    gsync();
    PROF_ADD(5, tick() - t1s_);
}

// ------------------------------------------------------------------------------------------------
// finish_direction: one pass that (1) evaluates the main part of G*dxi per row, (2) recovers the
// epigraph-variable steps, (3) completes gd = G*dxi, (4) recovers the multiplier steps dl.
// ------------------------------------------------------------------------------------------------
template <class M>
__device__ __forceinline__ void Ipm2<M>::finish_direction(double* w, double* rtil, double* rxv, double* dxi, double* gd, double* dl)
{
    const long long t0_ = tick();
    const double* socW = W + wo.socW;
    const double* nuv = W + wo.nuv;
    load_grows(L->g0, w); load_grows(L->g1, rtil);
    for (int i = lane; i < AG; i += 64) L->ga[i] = GAUX(rxv, i);
    if (lane < npa) L->pv[lane] = PV(dxi, lane);
    (void)socW;
    prefetch_r<S::O_D, SR>(0); pf_rows(pR0, w, 0); pf_rows(pR1, rtil, 0); pf_soc(0);
    pZ = Z(dxi, 0, lane < nz ? lane : nz - 1);
    pB1 = (lane < nz && N > 1) ? Z(dxi, 1, lane) : 0.0;
    pA = AUX(rxv, 0, lane < AS ? lane : AS - 1);
    pN = nuv[lane < MNU ? lane : MNU - 1];
    // per-node body; BND = boundary node (first / last): those two are peeled off so that the hot loop carries no
    // boundary-condition code and no node-type predicates
    auto node = [&](int k, auto bnd_tag) {
        constexpr bool BND = decltype(bnd_tag)::value;
        commit_r<S::O_D, SR>(); cm_rows(L->r0, pR0); cm_rows(L->r1, pR1); cm_soc();
        if (lane < nz) { L->zk[lane] = pZ; L->zn[lane] = pB1; }
        if (lane < AS) L->ak[lane] = pA;
        if (lane < MNU) L->nuk[lane] = pN;
        sync();
        if (k + 1 < N) {
            prefetch_r<S::O_D, SR>(k + 1); pf_rows(pR0, w, k + 1); pf_rows(pR1, rtil, k + 1); pf_soc(k + 1);
            pZ = pB1;
            pB1 = (lane < nz && k + 2 < N) ? Z(dxi, k + 2, lane) : 0.0;
            pA = AUX(rxv, k + 1, lane < AS ? lane : AS - 1);
            pN = nuv[(long)(k + 1) * MNU + (lane < MNU ? lane : MNU - 1)];
        }
        for (int r = lane; r < RS; r += 64) L->arow[r] = row_main(k, r);
        // boundary-condition rows (global) handled at their node
        if constexpr (BND) {
            const int nb = k == 0 ? nic : ntc;
            const double* H = k == 0 ? gH0() : gHf();
            const double* K = k == 0 ? gK0() : gKf();
            for (int i = lane; i < nb; i += 64) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < nx; j++) acc += H[i * nx + j] * L->zk[j];
#pragma unroll
                for (int j = 0; j < np; j++) acc += K[i * npa + j] * L->pv[j];
                L->tmp[i] = acc;
            }
        }
        sync();
        // ---- aux steps of this node ----
        if (lane < nx + ns) {
            const int c = lane;
            double val = 0.0;
            if (BND ? nu_live(k, c) : true) {   // mid nodes: every dynamics / hinge row is present
                double w1, w2, t1, t2, rxa; bool hg;
                nu_row_data(k, c, L->r0, L->r1, L->g0, L->g1, L->ak, L->ga, w1, w2, t1, t2, rxa, hg);
                const Pair pr = hg ? pairC(w1, w2, t1, t2, rxa) : pairA(w1, w2, t1, t2, rxa);
                const double av = c < nx ? L->arow[c] : L->arow[S::R_H0 + c - nx];
                val = (pr.rth + (hg ? w1 : (w1 - w2)) * av) / pr.Wt;
            }
            AUX(dxi, k, c) = val;
            L->thp[c] = val;   // aux step staged: [y (nx) | v (ns)]
        } else if (lane < nx + ns + 2) {
            const int which = lane - nx - ns;  // 0: eta_x, 1: eta_u
            const int j0 = which == 0 ? 0 : nx, n = which == 0 ? nx : nu;
            double Wt = 0.0, rth = -L->ak[which == 0 ? S::A_EX : S::A_EU], ha = 0.0;
            for (int q = 0; q < n; q++) {
                const double w1 = L->r0[S::R_TR0 + j0 + q], w2 = L->r0[S::R_TR1 + j0 + q];
                Wt += w1 + w2;
                rth += w1 * L->r1[S::R_TR0 + j0 + q] + w2 * L->r1[S::R_TR1 + j0 + q];
                ha += (w1 - w2) * L->zk[j0 + q];
            }
            const double val = (rth + ha) / Wt;
            AUX(dxi, k, which == 0 ? S::A_EX : S::A_EU) = val;
            L->thp[nx + ns + which] = val;
        }
        sync();
        // ---- gd and dl per row ----
        for (int r = lane; r < RS; r += 64) {
            double g, d;
            if (r < 2 * nx) {
                const int i = r % nx;
                if (BND ? (k < N - 1) : true) {
                    g = L->arow[r] - L->thp[i];
                    const double nv = L->nuk[i], rxa = L->ak[S::A_Y + i];
                    d = r < nx ? 0.5 * (rxa + nv) : 0.5 * (rxa - nv);
                } else { g = 0.0; d = 0.0; }
            } else if (r < S::R_TR0) {
                const int i = (r - S::R_H0) % (ns > 0 ? ns : 1);
                g = L->arow[r] - L->thp[nx + i];
                const double nv = L->nuk[nx + i], rxa = L->ak[S::A_V + i];
                d = r < S::R_H1 ? nv : rxa - nv;
            } else if (r < S::R_LIN) {
                const int j = (r - S::R_TR0) % nz;
                g = L->arow[r] - L->thp[nx + ns + (j < nx ? 0 : 1)];
                d = L->r0[r] * (g + L->r1[r]);
            } else if (r < S::R_SOC) {
                g = L->arow[r];
                d = L->r0[r] * (g + L->r1[r]);
            } else {
                g = L->arow[r];
                const int c = (r - S::R_SOC) / 4, rr = (r - S::R_SOC) % 4;
                const double* Wi = L->soc + c * 36 + 16;
                double t1[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    double acc = 0.0;
#pragma unroll
                    for (int q2 = 0; q2 < 4; q2++) acc += Wi[q * 4 + q2] * (L->arow[S::R_SOC + 4 * c + q2] + L->r1[S::R_SOC + 4 * c + q2]);
                    t1[q] = acc;
                }
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < 4; q++) acc += Wi[rr * 4 + q] * t1[q];
                d = acc;
            }
            ROW(gd, k, r) = g;
            ROW(dl, k, r) = d;
        }
        // boundary-condition rows of this node
        if constexpr (BND) {
            const bool isic = k == 0;
            const int nb = isic ? nic : ntc;
            for (int i = lane; i < nb; i += 64) {
                const int c = nx + ns + i;
                double w1, w2, t1, t2, rxa; bool hg;
                nu_row_data(k, c, L->r0, L->r1, L->g0, L->g1, L->ak, L->ga, w1, w2, t1, t2, rxa, hg);
                const Pair pr = pairA(w1, w2, t1, t2, rxa);
                const double av = L->tmp[i];
                const double dy = (pr.rth + (w1 - w2) * av) / pr.Wt;
                const double nv = L->nuk[c];
                GAUX(dxi, (isic ? S::GA_YIC : S::GA_YTC) + i) = dy;
                const int r0_ = isic ? S::G_IC0 : S::G_TC0, r1_ = isic ? S::G_IC1 : S::G_TC1;
                GROW(gd, r0_ + i) = av - dy; GROW(gd, r1_ + i) = -av - dy;
                GROW(dl, r0_ + i) = 0.5 * (rxa + nv); GROW(dl, r1_ + i) = 0.5 * (rxa - nv);
            }
        }
        sync();
        };
    node(0, std::true_type{});
#pragma unroll 1
    for (int k = 1; k < N - 1; k++) node(k, std::false_type{});
    if (N > 1) node(N - 1, std::true_type{});
    // ---- p trust region (global type B) and p-only rows ----
    if (lane == 0) {
        double detap = 0.0;
        if (np > 0) {
            double Wt = 0.0, rth = -L->ga[S::GA_EP], ha = 0.0;
            for (int q = 0; q < np; q++) {
                const double w1 = L->g0[S::G_TRP0 + q], w2 = L->g0[S::G_TRP1 + q];
                Wt += w1 + w2; rth += w1 * L->g1[S::G_TRP0 + q] + w2 * L->g1[S::G_TRP1 + q];
                ha += (w1 - w2) * L->pv[q];
            }
            detap = (rth + ha) / Wt;
        }
        GAUX(dxi, S::GA_EP) = detap;
        for (int j = 0; j < np; j++) {
            const double g0_ = L->pv[j] - detap, g1_ = -L->pv[j] - detap;
            GROW(gd, S::G_TRP0 + j) = g0_; GROW(gd, S::G_TRP1 + j) = g1_;
            GROW(dl, S::G_TRP0 + j) = L->g0[S::G_TRP0 + j] * (g0_ + L->g1[S::G_TRP0 + j]);
            GROW(dl, S::G_TRP1 + j) = L->g0[S::G_TRP1 + j] * (g1_ + L->g1[S::G_TRP1 + j]);
        }
        for (int i = 0; i < ng; i++) {
            double acc = 0.0;
            for (int j = 0; j < np; j++) acc += gLp()[i * npa + j] * L->pv[j];
            GROW(gd, S::G_LIN + i) = acc;
            GROW(dl, S::G_LIN + i) = L->g0[S::G_LIN + i] * (acc + L->g1[S::G_LIN + i]);
        }
    }
    gsync();
    PROF_ADD(6, tick() - t0_);
}

}  // namespace scp
