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
    // one v_mov_b64_dpp (gfx90a+: 64-bit DPP exists for row_newbcast); bound_ctrl with a zero `old` spares the destination's
    // initialisation (round 6; before: two 32-bit DPP moves, each behind a v_mov of the `old` value)
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + Q, 0xf, 0xf, true);
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
__device__ __forceinline__ bool chol_reg(const double* A, double* Lout, int lane, double (&a)[n], double (&d)[n])
{
    // a: row `lane` of A -> row of L (strictly lower part; a[j] of lane j: L_jj) ; d: 1 / L_jj (uniform)
    const int l16 = lane & 15;
    const int ln_ = l16 < n ? l16 : n - 1;   // unconditional (clamped) LDS reads: lanes >= n of a DPP row carry a copy of the last matrix row;
                                             // the four DPP rows of the wave hold identical copies (round 6: row 1 solves the arrow column)
#pragma unroll
    for (int c = 0; c < n; c++) a[c] = A[ln_ * ld + c];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < n; j++) {
        const double ajj = rl(a[j], j);
        ok = ok && (ajj > 0.0);
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
// Node-parallel part of the factorisation (round 6; before: inside the sequential node loop, 16 lanes live): everything of node k
// that does not depend on the chain X_{k-1} -- H0_k = Qd + L_inf blocks + Z' diag(omega) Z over the linear rows and the W^-1-scaled
// cone rows (-> workspace H0 [N][nz*nz]), C0_k (arrow), the per-row elimination coefficients (-> the Cf part of the factor
// records), the p x p term Dp (per-lane partial sums).  Four nodes per wave pass, one per DPP row of 16 lanes; lane sl holds column
// sl of H0 (symmetric), the outer products are row broadcasts.
template <class M>
__device__ __forceinline__ void Ipm2<M>::factor_pre(const double* w, double (&Dp)[npa * npa])
{
    double* H0g = W + wo.H0; double* gC0 = W + wo.C0;
    const double* socW = W + wo.socW;
    const int g = lane >> 4, sl = lane & 15;
    const int jc = sl < nz ? sl : nz - 1;
    const bool jx = jc < nx;
    constexpr int NZR = nl + 4 * nsoc;
    // per-lane constants of nu-row cl (see newton_rhs)
    const int cl = sl < MNU ? sl : MNU - 1;
    const bool c_isd = cl < nx, c_ish = (!c_isd) & (cl < nx + ns), c_bc = !(c_isd | c_ish);
    const int c_i = c_isd ? cl : (c_ish ? cl - nx : cl - nx - ns);
    const int c_oa = c_isd ? c_i : S::R_H0 + c_i, c_ob = c_isd ? nx + c_i : S::R_H1 + c_i;
    const int c_ga[2] = {N * RS + S::G_IC0 + c_i, N * RS + S::G_TC0 + c_i}, c_gb[2] = {N * RS + S::G_IC1 + c_i, N * RS + S::G_TC1 + c_i};
    // Branch-free body: lanes beyond a block and passes beyond the horizon repeat the last column / node and store the same values
    // again, so that the NB bodies of a loop iteration are one basic block and their loads overlap.
    auto body = [&](int k0) {
        const int k = k0 + g;
        const bool kv = k < N;
        const int kk = kv ? k : N - 1;
        const long rb = (long)kk * RS;
        const double* Pk_ = Pg + (long)kk * SR;
        // rows of Z (column jc) and their weights
        double z[NZR > 0 ? NZR : 1], om[NZR > 0 ? NZR : 1];
#pragma unroll
        for (int i = 0; i < nl; i++) { z[i] = Pk_[S::O_KL + (ns + i) * nz + jc]; om[i] = w[rb + S::R_LIN + i]; }
#pragma unroll
        for (int c = 0; c < nsoc; c++) {
            const double* Wi = socW + ((long)kk * nsoc + c) * 36 + 16;
            double kc[4];
#pragma unroll
            for (int q = 0; q < 4; q++) kc[q] = Pk_[S::O_KL + (ns + nl + 4 * c + q) * nz + jc];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < 4; q++) acc += Wi[rr * 4 + q] * kc[q];
                z[nl + 4 * c + rr] = acc; om[nl + 4 * c + rr] = 1.0;
            }
        }
        double h[nz];
#pragma unroll
        for (int a_ = 0; a_ < nz; a_++) h[a_] = 0.0;
#pragma unroll
        for (int r = 0; r < NZR; r++) {
            const double oz = om[r] * z[r];
#pragma unroll
            for (int a_ = 0; a_ < nz; a_++) h[a_] += oz * rl(z[r], a_);
        }
        // L_inf (trust-region) blocks: cancellation-free Schur complement entries (typeB_entry), group sums by row broadcast
        {
            const double w1 = w[rb + S::R_TR0 + jc], w2 = w[rb + S::R_TR1 + jc];
            const double d = w1 + w2, hb = w1 - w2;
            double Wt = 0.0, rest = 0.0;
#pragma unroll
            for (int q = 0; q < nz; q++) {
                const double dq = rl(d, q);
                const bool same = (q < nx) == jx;
                Wt += same ? dq : 0.0;
                rest += (same & (q != jc)) ? dq : 0.0;
            }
            const double ediag = 4.0 * w1 * w2 / d + hb * hb * rest / (d * Wt), hW = hb / Wt;
#pragma unroll
            for (int a_ = 0; a_ < nz; a_++) {
                const double ha = rl(hb, a_);
                const bool same = (a_ < nx) == jx;
                h[a_] += (a_ == jc) ? ediag : (same ? -ha * hW : 0.0);
            }
        }
        const double qd = Pk_[S::O_QD + jc];
#pragma unroll
        for (int a_ = 0; a_ < nz; a_++) H0g[(long)kk * HS + a_ * nz + jc] = h[a_] + ((a_ == jc) ? qd : 0.0);
        // arrow: C0_k[a = jc][j] and the p x p term
        if (np > 0) {
            double kp[nl > 0 ? nl : 1][npa];
#pragma unroll
            for (int i = 0; i < nl; i++)
#pragma unroll
                for (int j = 0; j < np; j++) kp[i][j] = Pk_[S::O_KP + (ns + i) * npa + j];
#pragma unroll
            for (int j = 0; j < np; j++) {
                double c0 = 0.0;
#pragma unroll
                for (int i = 0; i < nl; i++) c0 += om[i] * z[i] * kp[i][j];
                gC0[(long)kk * nz * npa + jc * npa + j] = c0;
            }
            const double once = (kv & (sl == 0)) ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < nl; i++)
#pragma unroll
                for (int p1 = 0; p1 < np; p1++)
#pragma unroll
                    for (int p2 = 0; p2 < np; p2++) Dp[p1 * npa + p2] += once * om[i] * kp[i][p1] * kp[i][p2];
        }
        // per-row elimination coefficients of nu-row cl (independent of the right-hand side)
        {
            const bool first = kk == 0, last = kk == N - 1, bndk = first | last;
            const int mmk = bndk ? MNU : MMID;
            const int c = cl < mmk ? cl : mmk - 1;   // (mid nodes have fewer rows: repeat the last one)
            const bool lv = (cl < mmk) ? (c_isd ? !last : (c_ish | (first & (c_i < nic)) | (last & (c_i < ntc)))) : true;
            const int rbi = kk * RS;
            const int gofs = first ? 0 : 1;
            int ia = c_bc ? c_ga[gofs] : rbi + c_oa;
            int ib = c_bc ? c_gb[gofs] : rbi + c_ob;
            const bool use = lv & (cl < mmk);
            ia = use ? ia : rbi; ib = use ? ib : rbi;
            double w1 = w[ia], w2 = w[ib];
            asm volatile("" : "+v"(w1), "+v"(w2));
            // lanes repeating row mmk - 1 must reproduce ITS values: its kind is dynamics / hinge (mid nodes), never a boundary row
            double cf0, cf1;
            if (cl < mmk) {
                w1 = lv ? w1 : 1.0; w2 = lv ? w2 : 1.0;
                const double iWt = fast_rcp(w1 + w2);
                const double kap = (c_ish ? 1.0 : 4.0) * w1 * w2 * iWt;
                cf0 = lv ? (c_ish ? w1 : (w1 - w2)) * iWt : 0.0;   // coefficient of rth in tau
                cf1 = lv ? fast_rcp(kap) : 1.0;                     // 1/kappa (1 for absent rows: identity pivot)
                double* cf = W + wo.F + (long)kk * FR + (bndk ? WK::f_cf(MNU) : WK::f_cf(MMID)) + c * 2;
                cf[0] = cf0; cf[1] = cf1;
            }
        }
    };
    constexpr int NB = 2;
#pragma unroll 1
    for (int k0 = 0; k0 < N; k0 += 4 * NB) {
#pragma unroll
        for (int u = 0; u < NB; u++) body(k0 + 4 * u);
    }
    gsync();
}

// ------------------------------------------------------------------------------------------------
// factor_stage: the CHAIN part of node k (everything else: factor_pre)
//   Sz_k  = H0_k + X_{k-1}' X_{k-1}      Lz = chol(Sz)          ("Li" in the record: the factor, reciprocal pivots on the diagonal)
//   Y_k   = Lz^-1 Dt_k'                  Snu_k = diag(1/kappa + reg) + Y'Y,  Ln = chol(Snu)   ("Lni")
//   X_k   = Ln^-1 Et_k                   (substitutions)
// plus the forward-substituted arrow columns.  pH: H0 in the MFMA accumulator layout, pCf: 1/kappa of row (lane & 15), pC0: C0 entry.
// ------------------------------------------------------------------------------------------------
template <class M>
template <int MM, int MP>
__device__ __forceinline__ void Ipm2<M>::factor_stage(int k, const double (&pH)[4], double pCf, double pC0)
{
    // MM: nu-rows of this node, MP: nu-rows of the previous node (rows of X_{k-1}; 0 at the first node) -- both compile time, so that
    // the interior-node instantiation <MMID, MMID> of the hot loop is straight-line code
    constexpr bool MID = (MM == MMID) && (MMID < MNU);   // instantiated for interior nodes only
    double* gYcz = W + wo.Ycz; double* gYcnu = W + wo.Ycnu;
    FPROF_BEGIN();
    // ---- Sz = H0_k + X'X  (X of the previous node is still in the factor record) ----
    // Matrix core: the rows of X_{k-1} are the K dimension of v_mfma_f64_16x16x4_f64 (A[i = lane & 15][k = lane >> 4] = X[r][i],
    // B[k][j = lane & 15] = X[r][j]; D row = (lane >> 4) + 4 reg, column = lane & 15), accumulated onto H0.
    {
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc4 = {pH[0], pH[1], pH[2], pH[3]};
        const int col = lane & 15, kq = lane >> 4, cc = col < nz ? col : nz - 1;
#pragma unroll
        for (int r0_ = 0; r0_ < MP; r0_ += 4) {
            const int r = r0_ + kq;
            double zb = Xm(MP > 0 ? MP : 1)[(r < MP ? r : (MP > 0 ? MP - 1 : 0)) * nz + cc];
            if (col >= nz || r >= MP) zb = 0.0;
            acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(zb, zb, acc4, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int a_ = kq + 4 * q, b_ = col;
            if (a_ < nz && b_ < nz) L->st.Sz[a_ * nz + b_] = acc4[q];
        }
    }
    // ---- forward substitution of the arrow columns: Cz = C0_k + X_{k-1}' ct_{k-1} ----
    if (np > 0) {
        for (int idx = lane; idx < nz * np; idx += 64) {
            const int a_ = idx / np, j = idx % np;
            double acc = pC0;
#pragma unroll
            for (int r = 0; r < MP; r++) acc += Xm(MP > 0 ? MP : 1)[r * nz + a_] * L->ct[r * npa + j];
            L->Cz[a_ * npa + j] = acc;
        }
    }
    sync();
    FPROF(0);
    // ---- Lz = chol(Sz) in registers (row per lane), written to the record as well ----
    double lz[nz], dz[nz];
    if (!chol_reg<nz, nz>(L->st.Sz, Li(), lane, lz, dz)) L->fail = 1;
    FPROF(1);
    // ---- Y = Lz^-1 Dt' : lane c owns column c (c < MM) ; lane MM (np > 0) does the arrow column cb = Lz^-1 Cz ----
    // forward substitution inside the lane; L[j][q] is broadcast from the registers of lane j (round 6; before: uniform LDS reads of
    // the packed factor behind a barrier)
    // lanes 16 .. 16 + np - 1 (DPP row 1, which holds its own copy of the factor) solve the arrow columns: a boundary node's block
    // fills row 0 (MNU = 16 for the rocket)
    static_assert(np <= 16, "arrow columns fit one DPP row");
    double y[nz];
    {
        const int l = lane & 15;
        const bool isC = (np > 0) && lane >= 16 && lane < 16 + np;
        const int jc_ = isC ? lane - 16 : 0;
        double dt[nz];
        if constexpr (MID) {
            const int lr = l < MM ? l : MM - 1;
            const double* src = isC ? L->Cz + jc_ : (lr < nx ? D() + lr * nz : Kl() + (lr - nx) * nz);
            const int st_ = isC ? npa : 1;
#pragma unroll
            for (int q = 0; q < nz; q++) dt[q] = src[q * st_];
        } else {
#pragma unroll
            for (int q = 0; q < nz; q++) dt[q] = isC ? L->Cz[q * npa + jc_] : Dt(k, l < MM ? l : MM - 1, q);
        }
#pragma unroll
        for (int j = 0; j < nz; j++) {
            double acc = dt[j];
#pragma unroll
            for (int q = 0; q < j; q++) acc -= rl(lz[q], j) * y[q];
            y[j] = acc * dz[j];
        }
        if (lane < MM) {
#pragma unroll
            for (int j = 0; j < nz; j++) Ym(MM)[j * MM + lane] = y[j];
        } else if (isC) {
#pragma unroll
            for (int j = 0; j < nz; j++) { L->cb[j * npa + jc_] = y[j]; gYcz[(long)k * nz * npa + j * npa + jc_] = y[j]; }
        }
    }
    FPROF(3);
    // arrow right-hand side Ft - Y' cb on lane q < MM (consumed by the X stage below); cb = column of lane MM (np == 1)
    double tq = 0.0;
    if constexpr (np == 1) {
        const int l = lane & 15;
        double v = (l < MM && nu_live_m<MID>(k, l < MM ? l : 0)) ? Ft_m<MID>(k, l < MM ? l : 0, 0) : 0.0;
#pragma unroll
        for (int i = 0; i < nz; i++) {
            const double cbi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y[i]), 16), __builtin_amdgcn_readlane(__double2loint(y[i]), 16));   // lane 16: the arrow column
            v -= y[i] * cbi;
        }
        tq = v;
    } else if constexpr (np > 1) {
        sync();
        for (int idx = lane; idx < MM * np; idx += 64) {
            const int q = idx / np, j = idx % np;
            double v = nu_live_m<MID>(k, q) ? Ft_m<MID>(k, q, j) : 0.0;
#pragma unroll
            for (int i = 0; i < nz; i++) v -= Ym(MM)[i * MM + q] * L->cb[i * npa + j];
            L->tmp[q * npa + j] = v;
        }
    }
    sync();
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
                if (c1 == c2) acc += pCf + (nu_live_m<MID>(k, c1) ? a.reg : 0.0);
                L->st.Snu[c1 * MNU + c2] = acc;
            }
        }
    }
    sync();
    FPROF(4);
    double ln[MM], dn[MM];
    if (!chol_reg<MM, MNU>(L->st.Snu, Lni(MM), lane, ln, dn)) L->fail = 1;
    FPROF(5);
    // ---- X = Ln^-1 Et : lane j owns column j (j < nz) ; arrow: ct = Ln^-1 (Ft - Y' cb) on lane nz (np == 1) ----
    {
        const int l = lane & 15;
        const bool isX = l < nz;
        double e[MM];
#pragma unroll
        for (int q = 0; q < MM; q++) {
            double v = (q < nx) ? E()[q * nz + (l < nz ? l : nz - 1)] : 0.0;
            if constexpr (np == 1) { const double tb = rl(tq, q); v = isX ? v : tb; }
            else if constexpr (np > 1) { if (!isX) v = (l < nz + np) ? L->tmp[q * npa + (l - nz)] : 0.0; }
            e[q] = v;
        }
        double x[MM];
#pragma unroll
        for (int c = 0; c < MM; c++) {
            double acc = e[c];
#pragma unroll
            for (int q = 0; q < c; q++) acc -= rl(ln[q], c) * x[q];
            x[c] = acc * dn[c];
        }
        sync();   // all reads of the previous X / ct are done before they are overwritten
        const bool isC = (np > 0) && lane >= nz && lane < nz + np;
        if (lane < nz) {
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
    for (int i = 0; i < npa * npa; i++) Dp[i] = 0.0;  // per-lane partial sums (factor_pre), reduced below
    load_grows(L->g0, w);
    (void)socW; (void)gC0; (void)gYcz; (void)gYcnu;
    gsync();
    factor_pre(w, Dp);
    if (np > 0) {
#pragma unroll
        for (int i = 0; i < npa * npa; i++) Dp[i] = wave_sum(Dp[i]);
    }
    // ---- chain: the first two and the last node are peeled off so that the hot loop holds a single instantiation ----
    const double* H0g = W + wo.H0; const double* Fg = W + wo.F;
    const int col_ = lane & 15, kq_ = lane >> 4;
    int offH[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int a_ = kq_ + 4 * q; offH[q] = (a_ < nz ? a_ : nz - 1) * nz + (col_ < nz ? col_ : nz - 1); }
    const int offCfM = WK::f_cf(MMID) + 2 * (col_ < MMID ? col_ : MMID - 1) + 1, offCfB = WK::f_cf(MNU) + 2 * (col_ < MNU ? col_ : MNU - 1) + 1;
    const int offC0 = lane < nz * np ? lane : (nz * np > 0 ? nz * np - 1 : 0);
    double pH[4], pCf, pC0, cH[4], cCf = 0.0, cC0 = 0.0;
    auto pf_pre = [&](int k) {
        const double* hb = H0g + (long)k * HS;
#pragma unroll
        for (int q = 0; q < 4; q++) pH[q] = hb[offH[q]];
        pCf = Fg[(long)k * FR + (bnd(k) ? offCfB : offCfM)];
        pC0 = np > 0 ? gC0[(long)k * nz * npa + offC0] : 0.0;
    };
    static_assert(nz * npa <= 64, "one C0 entry per lane");
    // the chain reads D, E, Fp and the HINGE rows of Kl / Kp only (the linear and cone rows went into H0 / C0 in factor_pre): half of a stage record
    constexpr int HI_ = S::O_KL + ns * nz;
    static_assert(ns * npa <= 64, "hinge rows of Kp: one load per lane");
    double pKp = 0.0;
    auto pf_kp = [&](int k) { if (ns > 0) pKp = Pg[(long)k * SR + S::O_KP + (lane < ns * npa ? lane : ns * npa - 1)]; };
    prefetch_r<S::O_D, HI_>(0); pf_kp(0); pf_pre(0);
    auto node_head = [&](int k) {
        commit_r<S::O_D, HI_>();
        if (ns > 0 && lane < ns * npa) L->st.Pk[S::O_KP + lane] = pKp;
#pragma unroll
        for (int q = 0; q < 4; q++) cH[q] = pH[q];
        cCf = pCf; cC0 = pC0;
        sync();
        if (k + 1 < N) { prefetch_r<S::O_D, HI_>(k + 1); pf_kp(k + 1); pf_pre(k + 1); }
    };
    node_head(0); factor_stage<MNU, 0>(0, cH, cCf, cC0); sync();
    if (N > 2) { node_head(1); factor_stage<MMID, MNU>(1, cH, cCf, cC0); sync(); }
#pragma unroll 1
    for (int k = 2; k < N - 1; k++) { node_head(k); factor_stage<MMID, MMID>(k, cH, cCf, cC0); sync(); }
    if (N > 2) { node_head(N - 1); factor_stage<MNU, MMID>(N - 1, cH, cCf, cC0); sync(); }
    else if (N > 1) { node_head(N - 1); factor_stage<MNU, MNU>(N - 1, cH, cCf, cC0); sync(); }
    gsync();
    // ---- arrow: back-substitute the np columns, then Sp = Dp0 - [C0; Ft]' Yc, chol(Sp) ----
    if (np > 0) {
        solve_backward_cols();
        double acc[npa * npa];
#pragma unroll
        for (int i = 0; i < npa * npa; i++) acc[i] = 0.0;
        arrow_dot<npa>(gYcz, gYcnu, acc);
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
        bwd_sweep(yz, yn, yz, yn);
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
// ---- chain sweeps on the padded per-lane factor record (Lds::Fpad) ----
// offset in Fpad of packed entry p of a factor record with MM nu-rows, for the forward (FWD) or the backward sweep; entries no
// sweep reads (the row coefficients, padding) go to per-lane dump slots
template <class M>
template <bool FWD, int MM>
__device__ __forceinline__ int Ipm2<M>::pad_off(int p) const
{
    constexpr int TZ = WK::tri(nz), TM = WK::tri(MM);
    auto row_of = [](int q, int n) { int i = 0; for (int t = 1; t < n; t++) i += (q >= t * (t + 1) / 2) ? 1 : 0; return i; };
    if (p < TZ) {
        const int i = row_of(p, nz), j = p - i * (i + 1) / 2;
        if (j == i) return FS_DA * 16 + i;
        return FWD ? (FS_A + j) * 16 + i : (FS_A + i) * 16 + j;
    }
    p -= TZ;
    if (p < TM) {
        const int i = row_of(p, MM), j = p - i * (i + 1) / 2;
        if (j == i) return FS_DC * 16 + i;
        return FWD ? (FS_C + j) * 16 + i : (FS_C + i) * 16 + j;
    }
    p -= TM;
    if (p < MM * nz) { const int r = p / nz, j = p % nz; return FWD ? (FS_D + r) * 16 + j : (FS_B + j) * 16 + r; }   // X[r][j]
    p -= MM * nz;
    if (p < nz * MM) { const int j = p / MM, c = p % MM; return FWD ? (FS_B + j) * 16 + c : (FS_D + c) * 16 + j; }   // Y[j][c]
    return FSLOTS * 16 + lane;
}
template <class M>
template <bool FWD>
__device__ __forceinline__ void Ipm2<M>::pad_begin(int (&offM)[NPREF_MID], int (&offB)[NPREF])
{
#pragma unroll
    for (int i = 0; i < NPREF_MID; i++) offM[i] = pad_off<FWD, MMID>(lane + 64 * i);
#pragma unroll
    for (int i = 0; i < NPREF; i++) offB[i] = pad_off<FWD, MNU>(lane + 64 * i);
    for (int i = lane; i < FS_RB * 16; i += 64) L->Fpad[i] = 0.0;   // the zeros of the triangles are never written again
    sync();
}
template <class M>
template <int NO, int NV>
__device__ __forceinline__ void Ipm2<M>::commit_pad(const int (&off)[NO], const double (&v)[NV], int i0)
{
#pragma unroll
    for (int i = 0; i < NV; i++) L->Fpad[off[i0 + i]] = v[i];
}
// Chain sweeps are bound by the latency of the loads that feed them (one wave per problem, nothing else to overlap with): the
// packed factor record and the two right-hand-side entries of a node are fetched PD nodes ahead into a register pipeline
// (round 6; before: one node ahead -- 2.6 us per node for 240 instructions).  The pipeline holds the INTERIOR nodes; its PD buffers
// are addressed statically (the node loop is unrolled PD times: a register move of a pending load would wait for it).  Node indices
// beyond the sweep are clamped (the load is issued, its value never used).  The two boundary nodes' records are longer (MNU
// rows) and are loaded on their own.
template <class M>
__device__ __forceinline__ void Ipm2<M>::chain_load(int k, double (&f)[NPREF_MID], double& b, double& t, const double* bv, const double* tv, int lz_, int lm_) const
{
    k = k < 0 ? 0 : (k > N - 1 ? N - 1 : k);
    const double* src = W + wo.F + (long)k * FR;
#pragma unroll
    for (int i = 0; i < NPREF_MID; i++) { const int idx = lane + 64 * i; f[i] = src[64 * (i + 1) <= FR ? idx : (idx < FR ? idx : FR - 1)]; }
    b = bv[(long)k * nz + lz_]; t = tv[(long)k * MNU + lm_];
}
template <class M>
__device__ __forceinline__ void Ipm2<M>::chain_load_bnd(int k, double (&f)[NPREF], double& b, double& t, const double* bv, const double* tv, int lz_, int lm_) const
{
    const double* src = W + wo.F + (long)k * FR;
#pragma unroll
    for (int i = 0; i < NPREF; i++) { const int idx = lane + 64 * i; f[i] = src[idx < FR ? idx : FR - 1]; }
    b = bv[(long)k * nz + lz_]; t = tv[(long)k * MNU + lm_];
}

// forward chain of node k: b-hat = Lz^-1 (b + X_{k-1}' t-hat_{k-1}) ; t-hat = Ln^-1 (t - Y' b-hat) ; returns X_k' t-hat
template <class M>
template <int MM>
__device__ __forceinline__ double Ipm2<M>::fwd_chain(int k, double znx)
{
    double* fb = W + wo.fb; double* ft = W + wo.ft;
    const int l = lane & 15;
    const double* P = L->Fpad + l;
    double li[nz], yc[nz], lni[MM], xc[MM];
#pragma unroll
    for (int q = 0; q < nz; q++) { li[q] = P[(FS_A + q) * 16]; yc[q] = P[(FS_B + q) * 16]; }
#pragma unroll
    for (int q = 0; q < MM; q++) { lni[q] = P[(FS_C + q) * 16]; xc[q] = P[(FS_D + q) * 16]; }
    const double dz_ = P[FS_DA * 16], dn_ = P[FS_DC * 16];
    const double b_in = P[FS_RB * 16], t_in = P[FS_RT * 16];
    const double b = l < nz ? b_in + znx : 0.0;
    const double bh = fsub16<nz>(b, li, dz_);
    double tp = l < MM ? t_in : 0.0;
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
// backward chain of node k: nu = Ln^-T (X z_{k+1} - t-hat) ; z = Lz^-T (b-hat - Y nu)
template <class M>
template <int MM>
__device__ __forceinline__ double Ipm2<M>::bwd_chain(int k, double zn, double* zo, double* nuo)
{
    const int l = lane & 15;
    const double* P = L->Fpad + l;
    double lic[nz], xr[nz], lnc[MM], yr[MM];
#pragma unroll
    for (int q = 0; q < nz; q++) { lic[q] = P[(FS_A + q) * 16]; xr[q] = P[(FS_B + q) * 16]; }
#pragma unroll
    for (int q = 0; q < MM; q++) { lnc[q] = P[(FS_C + q) * 16]; yr[q] = P[(FS_D + q) * 16]; }
    const double dz_ = P[FS_DA * 16], dn_ = P[FS_DC * 16];
    const double bh = (l < nz) ? P[FS_RB * 16] : 0.0;
    const double th = (l < MM) ? P[FS_RT * 16] : 0.0;
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
// backward sweep over the horizon: (b-hat, t-hat) in (bh_v [N][nz], th_v [N][MNU]) -> (zo, nuo); in place when zo == bh_v, nuo == th_v
template <class M>
__device__ __forceinline__ void Ipm2<M>::bwd_sweep(const double* bh_v, const double* th_v, double* zo, double* nuo)
{
    int offM[NPREF_MID], offB[NPREF];
    pad_begin<false>(offM, offB);
    const int lz_ = (lane & 15) < nz ? (lane & 15) : nz - 1, lm_ = (lane & 15) < MNU ? (lane & 15) : MNU - 1;
    double zn = 0.0;
    double fB[NPREF], bB, tB;
    double qf[PD][NPREF_MID], qb[PD], qt[PD];
    chain_load_bnd(N - 1, fB, bB, tB, bh_v, th_v, lz_, lm_);
#pragma unroll
    for (int d = 0; d < PD; d++) chain_load(N - 2 - d, qf[d], qb[d], qt[d], bh_v, th_v, lz_, lm_);
    auto bnd_node = [&](int k) {
        commit_pad(offB, fB, 0);
        L->Fpad[FS_RB * 16 + (lane & 15)] = bB; L->Fpad[FS_RT * 16 + (lane & 15)] = tB;
        sync();
        zn = bwd_chain<MNU>(k, zn, zo, nuo);
        sync();
    };
    auto mid = [&](int k, double (&f)[NPREF_MID], double& b, double& t) {
        commit_pad(offM, f, 0);
        L->Fpad[FS_RB * 16 + (lane & 15)] = b; L->Fpad[FS_RT * 16 + (lane & 15)] = t;   // (staged with the record: b, t are free for the next load)
        sync();
        chain_load(k - PD, f, b, t, bh_v, th_v, lz_, lm_);
        zn = bwd_chain<MMID>(k, zn, zo, nuo);
        sync();
    };
    bnd_node(N - 1);
    if (N > 1) chain_load_bnd(0, fB, bB, tB, bh_v, th_v, lz_, lm_);
    int k = N - 2;
#pragma unroll 1
    for (; k - PD >= 0; k -= PD) {
#pragma unroll
        for (int d = 0; d < PD; d++) mid(k - d, qf[d], qb[d], qt[d]);
    }
#pragma unroll
    for (int d = 0; d < PD - 1; d++) if (k - d >= 1) mid(k - d, qf[d], qb[d], qt[d]);
    if (N > 1) bnd_node(0);
    gsync();
}

// Right-hand sides of the forward chain for ALL nodes in one node-parallel pass (round 6; before: assembled inside the sequential
// node loop by the 11 + 9 lanes of the node's blocks, 2 000 instructions per node): b_k -> fb [N][nz], t_k -> ft [N][MNU], and the
// arrow right-hand side bp (per-lane partial sums).  Four nodes per wave pass, one per DPP row of 16 lanes: lane sl of a row holds
// component sl of b and row sl of t; the trust-region group sums are row broadcasts.
template <class M>
__device__ __forceinline__ void Ipm2<M>::newton_rhs(const double* w, const double* rtil, const double* rxv, double (&bp)[npa])
{
    double* fb = W + wo.fb; double* ft = W + wo.ft; double* tlv = W + wo.tl;
    const double* socW = W + wo.socW;
    // ---- cone rows: tl = W^-1 (W^-1 rtil), one cone per lane ----
    if (nsoc > 0) {
        for (int idx = lane; idx < N * nsoc; idx += 64) {
            const int k = idx / NSOC1, c = idx % NSOC1;
            const double* Wi = socW + (long)idx * 36 + 16;
            double wi[16], rt4[4], t1[4];
#pragma unroll
            for (int q = 0; q < 16; q++) wi[q] = Wi[q];
#pragma unroll
            for (int q = 0; q < 4; q++) rt4[q] = rtil[(long)k * RS + S::R_SOC + 4 * c + q];
#pragma unroll
            for (int r = 0; r < 4; r++) { double acc = 0.0; for (int q = 0; q < 4; q++) acc += wi[r * 4 + q] * rt4[q]; t1[r] = acc; }
#pragma unroll
            for (int r = 0; r < 4; r++) { double acc = 0.0; for (int q = 0; q < 4; q++) acc += wi[r * 4 + q] * t1[q]; tlv[(long)idx * 4 + r] = acc; }
        }
        gsync();
    }
    const int g = lane >> 4, sl = lane & 15;
    const int jc = sl < nz ? sl : nz - 1;
    const bool isx = sl < nx;
    const long NRS = (long)N * RS;
    // Branch-free body (every lane computes; lanes beyond a block and passes beyond the horizon repeat the last component / node and
    // store the same value again), so that the NB bodies of a loop iteration form one basic block and their loads overlap.
    const int cl = sl < MNU ? sl : MNU - 1;
    // per-lane constants of nu-row cl: dynamics (c_isd) / hinge (c_ish) / boundary-condition row (c_bc), offsets of its two rows in the
    // node's row record and of its aux entry; boundary-condition rows live in the global records ([0]: first node, [1]: last node)
    const bool c_isd = cl < nx, c_ish = (!c_isd) & (cl < nx + ns), c_bc = !(c_isd | c_ish);
    const int c_i = c_isd ? cl : (c_ish ? cl - nx : cl - nx - ns);
    const int c_oa = c_isd ? c_i : S::R_H0 + c_i, c_ob = c_isd ? nx + c_i : S::R_H1 + c_i, c_ox = c_isd ? S::A_Y + c_i : S::A_V + c_i;
    const int c_ga[2] = {N * RS + S::G_IC0 + c_i, N * RS + S::G_TC0 + c_i}, c_gb[2] = {N * RS + S::G_IC1 + c_i, N * RS + S::G_TC1 + c_i},
              c_gx[2] = {N * (nz + AS) + npa + S::GA_YIC + c_i, N * (nz + AS) + npa + S::GA_YTC + c_i};
    auto body = [&](int k0) {
        const int k = k0 + g;
        const bool kv = k < N;
        const int kk = kv ? k : N - 1;
        const long rb = (long)kk * RS;
        const double* Pk_ = Pg + (long)kk * SR;
        // ---- b_k[jc]: trust-region pair of component jc + group sums by row broadcast ----
        const double w1 = w[rb + S::R_TR0 + jc], w2 = w[rb + S::R_TR1 + jc], t1 = rtil[rb + S::R_TR0 + jc], t2 = rtil[rb + S::R_TR1 + jc];
        const double a1 = w1 + w2, a2 = w1 * t1 + w2 * t2;
        double Wx = 0.0, Wu = 0.0, rthx = -AUX((double*)rxv, kk, S::A_EX), rthu = -AUX((double*)rxv, kk, S::A_EU);
#pragma unroll
        for (int q = 0; q < nx; q++) { Wx += rl(a1, q); rthx += rl(a2, q); }
#pragma unroll
        for (int q = nx; q < nz; q++) { Wu += rl(a1, q); rthu += rl(a2, q); }
        const bool jx = jc < nx;
        double acc = -Z((double*)rxv, kk, jc);
        acc += -(w1 * t1 - w2 * t2) + (w1 - w2) * (jx ? rthx : rthu) * fast_rcp(jx ? Wx : Wu);
#pragma unroll
        for (int i = 0; i < nl; i++) acc += Pk_[S::O_KL + (ns + i) * nz + jc] * (-w[rb + S::R_LIN + i] * rtil[rb + S::R_LIN + i]);
#pragma unroll
        for (int r = 0; r < 4 * nsoc; r++) acc += Pk_[S::O_KL + (ns + nl + r) * nz + jc] * tlv[(long)kk * 4 * nsoc + r];
        fb[(long)kk * nz + jc] = acc;
        // ---- t_k[cl]: penalised (nu) row cl of the node ----
        {
            const int c = cl;
            const bool bndk = kk == 0 || kk == N - 1;
            const int mmk = bndk ? MNU : MMID;
            const bool first = kk == 0, last = kk == N - 1;
            // (32-bit indices from per-lane constants and non-short-circuit logic: selects, no branches)
            const bool lv = (c < mmk) & (c_isd ? !last : (c_ish | (first & (c_i < nic)) | (last & (c_i < ntc))));
            const int rbi = kk * RS, xa = N * nz + kk * AS;
            const int gofs = first ? 0 : 1;
            int ia = c_bc ? c_ga[gofs] : rbi + c_oa;
            int ib = c_bc ? c_gb[gofs] : rbi + c_ob;
            int ix = c_bc ? c_gx[gofs] : xa + c_ox;
            ia = lv ? ia : rbi; ib = lv ? ib : rbi; ix = lv ? ix : 0;
            const double v1 = w[ia], v2 = w[ib], u1 = rtil[ia], u2 = rtil[ib], rxa = rxv[ix];
            const double* cf = W + wo.F + (long)kk * FR + (bndk ? WK::f_cf(MNU) : WK::f_cf(MMID)) + (lv ? c : 0) * 2;
            const double cf0 = cf[0], cf1 = cf[1];
            const double r1 = v1 * u1, r2 = v2 * u2;
            const double rth = -rxa + r1 + r2;
            // tau = -(r1 - r2) + (w1-w2) rth / Wt  (type A)   |   -r1 + w1 rth / Wt  (hinge)
            const double tv = lv ? (-(r1 - (c_ish ? 0.0 : r2)) + cf0 * rth) * cf1 : 0.0;
            ft[(long)kk * MNU + c] = tv;
        }
        // ---- arrow right-hand side: bp += Kp' tl over the linear and cone rows ----
        if (np > 0) {
            constexpr int NZR = nl + 4 * nsoc;
#pragma unroll
            for (int r0_ = 0; r0_ < NZR; r0_ += 16) {
                const int r = r0_ + sl, rc = r < NZR ? r : NZR - 1;
                const bool lin = rc < nl;
                const long iw = rb + S::R_LIN + (lin ? rc : 0);
                const double tlin = -w[iw] * rtil[iw], tcone = tlv[(long)kk * 4 * nsoc + (lin ? 0 : rc - nl)];
                double tlin_ = tlin, tcone_ = tcone;
                asm volatile("" : "+v"(tlin_), "+v"(tcone_));   // (both loads unconditional: no branch around either)
                const double tl_ = (kv & (r < NZR)) ? (lin ? tlin_ : tcone_) : 0.0;
#pragma unroll
                for (int j = 0; j < np; j++) bp[j] += Pk_[S::O_KP + (ns + rc) * npa + j] * tl_;
            }
        }
    };
    constexpr int NB = 2;
#pragma unroll 1
    for (int k0 = 0; k0 < N; k0 += 4 * NB) {
#pragma unroll
        for (int u = 0; u < NB; u++) body(k0 + 4 * u);
    }
    gsync();
}

// acc[p1][p2] += sum over the horizon of [C0_k; Ft_k]' [yz_k; yn_k]: the arrow coupling of p with (z, nu).  yz: [N][nz][NC], yn: [N][MNU][NC]
// (NC = 1: one vector -- the Newton solves; NC = npa: the np back-substituted arrow columns -- the factorisation).  Flat over the
// (node, row) items of the z rows, the dynamics and the hinge rows (coefficients straight from the slab / the C0 workspace, two items per
// lane in flight, branch-free); the boundary-condition rows of the first and the last node come from the global record.  Per-lane
// partial sums: the caller reduces (round 6; before: a node loop through LDS with 27 of 64 lanes busy).
template <class M>
template <int NC>
__device__ __forceinline__ void Ipm2<M>::arrow_dot(const double* yz, const double* yn, double (&acc)[npa * NC])
{
    const double* gC0 = W + wo.C0;
    constexpr int NR = nz + nx + ns;
    const int cnt = N * NR;
#pragma unroll 1
    for (int base = 0; base < cnt; base += 128) {
        double cf_[2][npa], y_[2][NC];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int i0 = base + 64 * u + lane, idx = i0 < cnt ? i0 : cnt - 1;
            const int k = idx / NR, r = idx - k * NR, c = r - nz;
            const bool isz = r < nz, isd = c < nx;
            const double* cp = isz ? gC0 + ((long)k * nz + r) * npa : Pg + (long)k * SR + (isd ? S::O_FP + c * npa : S::O_KP + (c - nx) * npa);
            const double* yp = isz ? yz + ((long)k * nz + r) * NC : yn + ((long)k * MNU + c) * NC;
            const double f = (i0 < cnt && (isz || !isd || k < N - 1)) ? 1.0 : 0.0;   // (no dynamics rows at the last node)
#pragma unroll
            for (int j = 0; j < np; j++) cf_[u][j] = f * cp[j];
#pragma unroll
            for (int q = 0; q < NC; q++) y_[u][q] = yp[q];
        }
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int p1 = 0; p1 < np; p1++)
#pragma unroll
                for (int q = 0; q < NC; q++) acc[p1 * NC + q] += cf_[u][p1] * y_[u][q];
    }
    for (int which = 0; which < 2; which++) {
        const int k = which == 0 ? 0 : N - 1, nb = which == 0 ? nic : ntc;
        const double* K = which == 0 ? gK0() : gKf();
        for (int i = lane; i < nb; i += 64) {
            const double* yp = yn + ((long)k * MNU + nx + ns + i) * NC;
#pragma unroll
            for (int p1 = 0; p1 < np; p1++)
#pragma unroll
                for (int q = 0; q < NC; q++) acc[p1 * NC + q] += K[i * npa + p1] * yp[q];
        }
    }
}

template <class M>
__device__ __forceinline__ void Ipm2<M>::newton_solve(double* w, double* rtil, double* rxv, double* dxi)
{
    const long long t0s_ = tick();
    double* fb = W + wo.fb; double* ft = W + wo.ft; double* nuv = W + wo.nuv;
    double bp[npa];
#pragma unroll
    for (int j = 0; j < npa; j++) bp[j] = 0.0;
    load_grows(L->g0, w); load_grows(L->g1, rtil);
    for (int i = lane; i < AG; i += 64) L->ga[i] = GAUX(rxv, i);
    gsync();
    newton_rhs(w, rtil, rxv, bp);
    // ---------------- forward sweep ----------------
    {
        int offM[NPREF_MID], offB[NPREF];
        pad_begin<true>(offM, offB);
        const int lz_ = (lane & 15) < nz ? (lane & 15) : nz - 1, lm_ = (lane & 15) < MNU ? (lane & 15) : MNU - 1;
        double znx = 0.0;   // lane j: (X_{k-1}' t-hat_{k-1})_j
        double fB[NPREF], bB, tB;
        double qf[PD][NPREF_MID], qb[PD], qt[PD];
        chain_load_bnd(0, fB, bB, tB, fb, ft, lz_, lm_);
#pragma unroll
        for (int d = 0; d < PD; d++) chain_load(1 + d, qf[d], qb[d], qt[d], fb, ft, lz_, lm_);
        // boundary nodes peeled off: the hot loop holds the mid-node instantiation only
        auto bnd_node = [&](int k) {
            commit_pad(offB, fB, 0);
            L->Fpad[FS_RB * 16 + (lane & 15)] = bB; L->Fpad[FS_RT * 16 + (lane & 15)] = tB;
            sync();
            znx = fwd_chain<MNU>(k, znx);
            sync();
        };
        auto mid = [&](int k, double (&f)[NPREF_MID], double& b, double& t) {
            commit_pad(offM, f, 0);
            L->Fpad[FS_RB * 16 + (lane & 15)] = b; L->Fpad[FS_RT * 16 + (lane & 15)] = t;   // (staged with the record: b, t are free for the next load)
            sync();
            chain_load(k + PD, f, b, t, fb, ft, lz_, lm_);
            znx = fwd_chain<MMID>(k, znx);
            sync();
        };
        bnd_node(0);
        if (N > 1) chain_load_bnd(N - 1, fB, bB, tB, fb, ft, lz_, lm_);
        int k = 1;
#pragma unroll 1
        for (; k + PD <= N - 1; k += PD) {
#pragma unroll
            for (int d = 0; d < PD; d++) mid(k + d, qf[d], qb[d], qt[d]);
        }
#pragma unroll
        for (int d = 0; d < PD - 1; d++) if (k + d < N - 1) mid(k + d, qf[d], qb[d], qt[d]);
        if (N > 1) bnd_node(N - 1);
        gsync();
    }
    PROF_ADD(3, tick() - t0s_);
    const long long tb_ = tick();
    // ---------------- backward sweep ----------------
    bwd_sweep(fb, ft, dxi, nuv);
    PROF_ADD(4, tick() - tb_);
    const long long t1s_ = tick();
    // ---------------- arrow: dp = Sp^-1 (bp - [C0; Ft]' y_b) ; z -= Ycz dp ; nu -= Ycnu dp ----------------
    double dp[npa];
#pragma unroll
    for (int j = 0; j < npa; j++) dp[j] = 0.0;
    if (np > 0) {
        const double* gC0 = W + wo.C0; const double* gYcz = W + wo.Ycz; const double* gYcnu = W + wo.Ycnu;
        (void)gYcz; (void)gYcnu;
        {
            double acc1[npa];
#pragma unroll
            for (int j = 0; j < npa; j++) acc1[j] = 0.0;
            arrow_dot<1>(dxi, nuv, acc1);
#pragma unroll
            for (int j = 0; j < np; j++) bp[j] -= acc1[j];
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
    // Round 6: FLAT by row type like G_apply (before: one node at a time through LDS, 930 instructions and an exposed memory round
    // trip per node): every item reads what it needs straight from global memory, two items per lane in flight, stores last.
    const long long t0_ = tick();
    const double* socW = W + wo.socW;
    const double* nuv = W + wo.nuv;
    load_grows(L->g0, w); load_grows(L->g1, rtil);
    for (int i = lane; i < AG; i += 64) L->ga[i] = GAUX(rxv, i);
    if (lane < npa) L->pv[lane] = PV(dxi, lane);
    double pv_[npa];
#pragma unroll
    for (int j = 0; j < npa; j++) pv_[j] = np > 0 ? PV(dxi, j) : 0.0;
    sync();
    // ---- dynamics rows: step of the L1 epigraph variable y, gd and dl of the row pair ----
    {
        struct R { double *a, *g0, *g1, *d0, *d1; double va, vg0, vg1, vd0, vd1;
                   __device__ __forceinline__ void store() const { *a = va; *g0 = vg0; *g1 = vg1; *d0 = vd0; *d1 = vd1; } };
        flat_items(N * nx, [&](int idx) {
            const int k = idx / nx, i = idx - k * nx, kn = k + 1 < N ? k + 1 : N - 1;
            const double* Pk_ = Pg + (long)k * SR;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < nz; j++) acc += Pk_[S::O_D + i * nz + j] * Z(dxi, k, j) + Pk_[S::O_E + i * nz + j] * Z(dxi, kn, j);
#pragma unroll
            for (int j = 0; j < np; j++) acc += Pk_[S::O_FP + i * npa + j] * pv_[j];
            const double w1 = ROW(w, k, i), w2 = ROW(w, k, nx + i), t1 = ROW(rtil, k, i), t2 = ROW(rtil, k, nx + i);
            const double rxa = AUX(rxv, k, S::A_Y + i), nv = nuv[(long)k * MNU + i];
            const Pair pr = pairA(w1, w2, t1, t2, rxa);
            const bool lv = k < N - 1;
            const double val = lv ? (pr.rth + (w1 - w2) * acc) / pr.Wt : 0.0;
            return R{&AUX(dxi, k, i), &ROW(gd, k, i), &ROW(gd, k, nx + i), &ROW(dl, k, i), &ROW(dl, k, nx + i),
                     val, lv ? acc - val : 0.0, lv ? -acc - val : 0.0, lv ? 0.5 * (rxa + nv) : 0.0, lv ? 0.5 * (rxa - nv) : 0.0};
        });
    }
    // ---- hinge rows: step of the hinge epigraph variable v ----
    if (ns > 0) {
        struct R { double *a, *g0, *g1, *d0, *d1; double va, vg0, vg1, vd0, vd1;
                   __device__ __forceinline__ void store() const { *a = va; *g0 = vg0; *g1 = vg1; *d0 = vd0; *d1 = vd1; } };
        flat_items(N * ns, [&](int idx) {
            const int k = idx / (ns > 0 ? ns : 1), i = idx - k * ns;
            const double* Pk_ = Pg + (long)k * SR;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < nz; j++) acc += Pk_[S::O_KL + i * nz + j] * Z(dxi, k, j);
#pragma unroll
            for (int j = 0; j < np; j++) acc += Pk_[S::O_KP + i * npa + j] * pv_[j];
            const double w1 = ROW(w, k, S::R_H0 + i), w2 = ROW(w, k, S::R_H1 + i), t1 = ROW(rtil, k, S::R_H0 + i), t2 = ROW(rtil, k, S::R_H1 + i);
            const double rxa = AUX(rxv, k, S::A_V + i), nv = nuv[(long)k * MNU + nx + i];
            const Pair pr = pairC(w1, w2, t1, t2, rxa);
            const double val = (pr.rth + w1 * acc) / pr.Wt;
            return R{&AUX(dxi, k, nx + i), &ROW(gd, k, S::R_H0 + i), &ROW(gd, k, S::R_H1 + i), &ROW(dl, k, S::R_H0 + i), &ROW(dl, k, S::R_H1 + i),
                     val, acc - val, 0.0 - val, nv, rxa - nv};
        });
    }
    // ---- linear rows ----
    if (nl > 0) {
        struct R { double *g0, *d0; double vg0, vd0; __device__ __forceinline__ void store() const { *g0 = vg0; *d0 = vd0; } };
        flat_items(N * nl, [&](int idx) {
            const int k = idx / (nl > 0 ? nl : 1), i = idx - k * nl;
            const double* Pk_ = Pg + (long)k * SR;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < nz; j++) acc += Pk_[S::O_KL + (ns + i) * nz + j] * Z(dxi, k, j);
#pragma unroll
            for (int j = 0; j < np; j++) acc += Pk_[S::O_KP + (ns + i) * npa + j] * pv_[j];
            const int r = S::R_LIN + i;
            return R{&ROW(gd, k, r), &ROW(dl, k, r), acc, ROW(w, k, r) * (acc + ROW(rtil, k, r))};
        });
    }
    // ---- cone rows: gd = -(K z + Kp p), dl = W^-1 (W^-1 (gd + rtil)), one cone per item ----
    if (nsoc > 0) {
        struct R { double *g, *d; double vg[4], vd[4];
                   __device__ __forceinline__ void store() const { for (int q = 0; q < 4; q++) { g[q] = vg[q]; d[q] = vd[q]; } } };
        flat_items(N * nsoc, [&](int idx) {
            const int k = idx / NSOC1, c = idx - k * NSOC1;
            const double* Pk_ = Pg + (long)k * SR;
            const double* Wi = socW + (long)idx * 36 + 16;
            R out_;
            double u[4], t1[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = ns + nl + 4 * c + q;
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < nz; j++) acc += Pk_[S::O_KL + row * nz + j] * Z(dxi, k, j);
#pragma unroll
                for (int j = 0; j < np; j++) acc += Pk_[S::O_KP + row * npa + j] * pv_[j];
                out_.vg[q] = -acc;
                u[q] = -acc + ROW(rtil, k, S::R_SOC + 4 * c + q);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) { double acc = 0.0; for (int q2 = 0; q2 < 4; q2++) acc += Wi[q * 4 + q2] * u[q2]; t1[q] = acc; }
#pragma unroll
            for (int q = 0; q < 4; q++) { double acc = 0.0; for (int q2 = 0; q2 < 4; q2++) acc += Wi[q * 4 + q2] * t1[q2]; out_.vd[q] = acc; }
            out_.g = &ROW(gd, k, S::R_SOC + 4 * c); out_.d = &ROW(dl, k, S::R_SOC + 4 * c);
            return out_;
        });
    }
    // ---- trust-region rows: steps of eta_x, eta_u (group sums), gd and dl of the row pairs; item = (node, group) ----
    {
        struct R { double *a, *g0, *g1, *d0, *d1; double va; double vg0[nx > nu ? nx : nu], vg1[nx > nu ? nx : nu], vd0[nx > nu ? nx : nu], vd1[nx > nu ? nx : nu]; int n;
                   __device__ __forceinline__ void store() const {
                       *a = va;
#pragma unroll
                       for (int q = 0; q < (nx > nu ? nx : nu); q++) if (q < n) { g0[q] = vg0[q]; g1[q] = vg1[q]; d0[q] = vd0[q]; d1[q] = vd1[q]; }
                   } };
        flat_items(N * 2, [&](int idx) {
            const int k = idx >> 1, which = idx & 1;
            const int j0 = which == 0 ? 0 : nx, n = which == 0 ? nx : nu;
            constexpr int NM = nx > nu ? nx : nu;
            R o_;
            double w1[NM], w2[NM], t1[NM], t2[NM], zj[NM];
            double Wt = 0.0, rth = -AUX(rxv, k, which == 0 ? S::A_EX : S::A_EU), ha = 0.0;
#pragma unroll
            for (int q = 0; q < NM; q++) {
                const int jq = j0 + (q < n ? q : n - 1);
                w1[q] = ROW(w, k, S::R_TR0 + jq); w2[q] = ROW(w, k, S::R_TR1 + jq);
                t1[q] = ROW(rtil, k, S::R_TR0 + jq); t2[q] = ROW(rtil, k, S::R_TR1 + jq);
                zj[q] = Z(dxi, k, jq);
                if (q < n) { Wt += w1[q] + w2[q]; rth += w1[q] * t1[q] + w2[q] * t2[q]; ha += (w1[q] - w2[q]) * zj[q]; }
            }
            const double val = (rth + ha) / Wt;
#pragma unroll
            for (int q = 0; q < NM; q++) {
                const double g0 = zj[q] - val, g1 = -zj[q] - val;
                o_.vg0[q] = g0; o_.vg1[q] = g1; o_.vd0[q] = w1[q] * (g0 + t1[q]); o_.vd1[q] = w2[q] * (g1 + t2[q]);
            }
            o_.va = val; o_.n = n;
            o_.a = &AUX(dxi, k, which == 0 ? S::A_EX : S::A_EU);
            o_.g0 = &ROW(gd, k, S::R_TR0 + j0); o_.g1 = &ROW(gd, k, S::R_TR1 + j0); o_.d0 = &ROW(dl, k, S::R_TR0 + j0); o_.d1 = &ROW(dl, k, S::R_TR1 + j0);
            return o_;
        });
    }
    // ---- boundary-condition rows (global records) of the first and the last node ----
    for (int which = 0; which < 2; which++) {
        const bool isic = which == 0;
        const int k = isic ? 0 : N - 1;
        const int nb = isic ? nic : ntc;
        const double* H = isic ? gH0() : gHf();
        const double* K = isic ? gK0() : gKf();
        for (int i = lane; i < nb; i += 64) {
            double av = 0.0;
#pragma unroll
            for (int j = 0; j < nx; j++) av += H[i * nx + j] * Z(dxi, k, j);
#pragma unroll
            for (int j = 0; j < np; j++) av += K[i * npa + j] * pv_[j];
            const int c = nx + ns + i;
            double w1, w2, t1, t2, rxa; bool hg;
            nu_row_data(k, c, L->r0, L->r1, L->g0, L->g1, L->ak, L->ga, w1, w2, t1, t2, rxa, hg);
            const Pair pr = pairA(w1, w2, t1, t2, rxa);
            const double dy = (pr.rth + (w1 - w2) * av) / pr.Wt;
            const double nv = nuv[(long)k * MNU + c];
            GAUX(dxi, (isic ? S::GA_YIC : S::GA_YTC) + i) = dy;
            const int r0_ = isic ? S::G_IC0 : S::G_TC0, r1_ = isic ? S::G_IC1 : S::G_TC1;
            GROW(gd, r0_ + i) = av - dy; GROW(gd, r1_ + i) = -av - dy;
            GROW(dl, r0_ + i) = 0.5 * (rxa + nv); GROW(dl, r1_ + i) = 0.5 * (rxa - nv);
        }
    }
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
