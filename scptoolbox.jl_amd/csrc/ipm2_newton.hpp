// Newton system of the structured IPM (v2): factorisation sweep, solve sweeps, direction recovery.
// Included by ipm2_kernel.hpp.  Algebra: oracle/ipm_struct.py (qd_factor / qd_solve / newton).
#pragma once

namespace scp {

// ------------------------------------------------------------------------------------------------
// factor: forward sweep over the nodes.
//   Sz_k  = H0_k + X_{k-1}' X_{k-1}      Lz = chol(Sz),  Li = Lz^-1
//   Y_k   = Li Dt_k'                     Snu_k = diag(1/kappa + reg) + Y'Y,  Lnu = chol, Lni = Lnu^-1
//   X_k   = Lni Et_k
// plus the forward-substituted arrow columns (C0_k / Ft_k) and the np x np Schur complement.
// ------------------------------------------------------------------------------------------------
template <class M>
__device__ __forceinline__ void Ipm2<M>::factor(double* w)
{
    const long long t0_ = tick();
    double* gC0 = W + wo.C0; double* gYcz = W + wo.Ycz; double* gYcnu = W + wo.Ycnu;
    const double* socW = W + wo.socW;
    double Dp[npa * npa];
#pragma unroll
    for (int i = 0; i < npa * npa; i++) Dp[i] = 0.0;  // lane 0 accumulates
    load_grows(L->g0, w);
    prefetch(0);
    for (int k = 0; k < N; k++) {
        const int m = mnu(k);
        commit();
        load_rows(L->r0, w, k);
        for (int i = lane; i < nsoc * 36; i += 64) L->soc[i] = socW[(long)k * nsoc * 36 + i];
        sync();
        if (k + 1 < N) prefetch(k + 1);
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
        // ---- Sz = H0_k + X'X ----
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
                const int mp = mnu(k - 1);
                for (int r = 0; r < mp; r++) acc += Xm()[r * nz + a_] * Xm()[r * nz + b_];
            }
            L->Sz[idx] = acc;
        }
        // ---- C0_k (arrow coupling of the local rows) ----
        for (int idx = lane; idx < nz * npa; idx += 64) {
            const int a_ = idx / npa, j = idx % npa;
            double acc = 0.0;
            if (np > 0) {
#pragma unroll
                for (int i = 0; i < nl; i++) acc += L->r0[S::R_LIN + i] * Kl()[(ns + i) * nz + a_] * Kp()[(ns + i) * npa + j];
                // + X_{k-1}' t-hat_{k-1} columns (forward substitution of the arrow columns)
                if (k > 0) {
                    const int mp = mnu(k - 1);
                    for (int r = 0; r < mp; r++) acc += Xm()[r * nz + a_] * L->ct[r * npa + j];
                }
            }
            L->Cz[idx] = acc;
        }
        if (np > 0) {
            // C0 itself (without the forward term) is needed again by the p-system
            for (int idx = lane; idx < nz * npa; idx += 64) {
                const int a_ = idx / npa, j = idx % npa;
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < nl; i++) acc += L->r0[S::R_LIN + i] * Kl()[(ns + i) * nz + a_] * Kp()[(ns + i) * npa + j];
                gC0[(long)k * nz * npa + idx] = acc;
            }
            if (lane == 0)
                for (int i = 0; i < nl; i++)
                    for (int p1 = 0; p1 < np; p1++)
                        for (int p2 = 0; p2 < np; p2++)
                            Dp[p1 * npa + p2] += L->r0[S::R_LIN + i] * Kp()[(ns + i) * npa + p1] * Kp()[(ns + i) * npa + p2];
        }
        sync();
        chol<nz, nz>(L->Sz);
        tri_inverse<nz, nz>(L->Sz, Li());
        // ---- Y = Li Dt' (nz x m) ; cb = Li Cz ----
        for (int idx = lane; idx < nz * m; idx += 64) {
            const int j = idx / m, c = idx % m;
            double acc = 0.0;
            for (int q = 0; q <= j; q++) acc += Li()[j * nz + q] * Dt(k, c, q);
            Ym()[j * MNU + c] = acc;
        }
        if (np > 0)
            for (int idx = lane; idx < nz * np; idx += 64) {
                const int j = idx / np, c = idx % np;
                double acc = 0.0;
                for (int q = 0; q <= j; q++) acc += Li()[j * nz + q] * L->Cz[q * npa + c];
                L->cb[j * npa + c] = acc;
            }
        sync();
        // ---- Snu = diag(kinv + reg) + Y'Y ----
        for (int idx = lane; idx < m * m; idx += 64) {
            const int c1 = idx / m, c2 = idx % m;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < nz; j++) acc += Ym()[j * MNU + c1] * Ym()[j * MNU + c2];
            if (c1 == c2) {
                double ki = 1.0;
                const bool lv = nu_live(k, c1);
                if (lv) {
                    double w1, w2, t1, t2, rxa; bool hg;
                    nu_row_data(k, c1, L->r0, L->r0, L->g0, L->g0, L->ak, L->ga, w1, w2, t1, t2, rxa, hg);
                    ki = hg ? (w1 + w2) / (w1 * w2) : (w1 + w2) / (4.0 * w1 * w2);
                }
                acc += ki + (lv ? a.reg : 0.0);
            }
            L->Snu[c1 * MNU + c2] = acc;
        }
        sync();
        if (m == MNU) { chol<MNU, MNU>(L->Snu); tri_inverse<MNU, MNU>(L->Snu, Lni()); }
        else { chol<MMID, MNU>(L->Snu); tri_inverse<MMID, MNU>(L->Snu, Lni()); }
        // ---- ct = Lni (Ft - Y' cb) (arrow columns) ; X = Lni Et (only the dyn rows of Et are non-zero) ----
        if (np > 0) {
            for (int idx = lane; idx < m * np; idx += 64) {
                const int c = idx / np, j = idx % np;
                double acc = nu_live(k, c) ? Ft(k, c, j) : 0.0;
#pragma unroll
                for (int q = 0; q < nz; q++) acc -= Ym()[q * MNU + c] * L->cb[q * npa + j];
                L->tmp[c * npa + j] = acc;   // MNU * npa <= 64 is asserted in run()
            }
            sync();
            for (int idx = lane; idx < m * np; idx += 64) {
                const int c = idx / np, j = idx % np;
                double acc = 0.0;
                for (int q = 0; q <= c; q++) acc += Lni()[c * MNU + q] * L->tmp[q * npa + j];
                L->ct[c * npa + j] = acc;
            }
            for (int idx = lane; idx < nz * np; idx += 64) gYcz[(long)k * nz * npa + idx] = L->cb[(idx / np) * npa + idx % np];
        }
        for (int idx = lane; idx < m * nz; idx += 64) {
            const int c = idx / nz, j = idx % nz;
            double acc = 0.0;
            const int qmax = c < nx ? c : nx - 1;   // Et rows >= nx are zero
#pragma unroll 1
            for (int q = 0; q <= qmax; q++) acc += Lni()[c * MNU + q] * E()[q * nz + j];
            Xm()[c * nz + j] = acc;
        }
        sync();
        if (np > 0)
            for (int idx = lane; idx < m * np; idx += 64) gYcnu[(long)k * MNU * npa + (idx / np) * npa + idx % np] = L->ct[(idx / np) * npa + idx % np];
        storeF(k);
        sync();
    }
    // ---- arrow: back-substitute the np columns, then Sp = Dp0 - [C0; Ft]' Yc, chol(Sp) ----
    if (np > 0) {
        solve_backward_cols();
        double acc[npa * npa];
#pragma unroll
        for (int i = 0; i < npa * npa; i++) acc[i] = 0.0;
        prefetch(0);
        for (int k = 0; k < N; k++) {
            commit();
            sync();
            if (k + 1 < N) prefetch(k + 1);
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
#pragma unroll
        for (int i = 0; i < npa * npa; i++) spL[i] = L->tmp[i];
        sync();
    }
    prof[2] += tick() - t0_;
}

// backward sweep for the np arrow columns held in (Ycz = b-hat, Ycnu = t-hat)
template <class M>
__device__ __forceinline__ void Ipm2<M>::solve_backward_cols()
{
    double* yz = W + wo.Ycz; double* yn = W + wo.Ycnu;
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
                for (int q = 0; q < nz; q++) acc += Xm()[c * nz + q] * L->Cz[q * npa + j];   // Cz holds z_{k+1} columns
            }
            L->tmp[c * npa + j] = acc;
        }
        sync();
        // nu = Lni' u
        for (int idx = lane; idx < m * np; idx += 64) {
            const int c = idx / np, j = idx % np;
            double acc = 0.0;
            for (int r = c; r < m; r++) acc += Lni()[r * MNU + c] * L->tmp[r * npa + j];
            L->ct[c * npa + j] = acc;
        }
        sync();
        for (int idx = lane; idx < m * np; idx += 64) yn[(long)k * MNU * npa + (idx / np) * npa + idx % np] = L->ct[(idx / np) * npa + idx % np];
        // v = b-hat - Y nu ; z = Li' v
        for (int idx = lane; idx < nz * np; idx += 64) {
            const int q = idx / np, j = idx % np;
            double acc = L->cb[q * npa + j];
            for (int c = 0; c < m; c++) acc -= Ym()[q * MNU + c] * L->ct[c * npa + j];
            L->tmp[q * npa + j] = acc;
        }
        sync();
        for (int idx = lane; idx < nz * np; idx += 64) {
            const int q = idx / np, j = idx % np;
            double acc = 0.0;
#pragma unroll 1
            for (int r = q; r < nz; r++) acc += Li()[r * nz + q] * L->tmp[r * npa + j];
            L->Cz[q * npa + j] = acc;
            yz[(long)k * nz * npa + q * npa + j] = acc;
        }
        sync();
    }
}

// ------------------------------------------------------------------------------------------------
// newton_solve: (P + G'W^-2 G) dxi = -rxv - G'W^-2 rtil with the stored factorisation.
// Writes the MAIN part of dxi (dz, dp) and nu; finish_direction() completes aux / dlam.
// ------------------------------------------------------------------------------------------------
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
    if (lane < nz) L->znx[lane] = 0.0;   // X_{k-1}' t-hat_{k-1}
    prefetch(0); prefetchF(0);
    for (int k = 0; k < N; k++) {
        const int m = mnu(k);
        commit(); commitF();
        load_rows(L->r0, w, k); load_rows(L->r1, rtil, k);
        if (lane < nz) L->zk[lane] = Z(rxv, k, lane);
        if (lane < AS) L->ak[lane] = AUX(rxv, k, lane);
        for (int i = lane; i < nsoc * 36; i += 64) L->soc[i] = socW[(long)k * nsoc * 36 + i];
        sync();
        if (k + 1 < N) { prefetch(k + 1); prefetchF(k + 1); }
        // cone rows: tl = W^-1 (W^-1 rtil)
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
        sync();
        // right-hand sides b_k (+ X' t-hat of the previous node), t_k
        if (lane < nz) {
            const int j = lane;
            double acc = -L->zk[j] + L->znx[j];
            const int j0 = j < nx ? 0 : nx, n = j < nx ? nx : nu;
            double Wt = 0.0, rth = -L->ak[j < nx ? S::A_EX : S::A_EU];
            for (int q = 0; q < n; q++) {
                const double w1 = L->r0[S::R_TR0 + j0 + q], w2 = L->r0[S::R_TR1 + j0 + q];
                Wt += w1 + w2;
                rth += w1 * L->r1[S::R_TR0 + j0 + q] + w2 * L->r1[S::R_TR1 + j0 + q];
            }
            const double w1 = L->r0[S::R_TR0 + j], w2 = L->r0[S::R_TR1 + j];
            acc += -(w1 * L->r1[S::R_TR0 + j] - w2 * L->r1[S::R_TR1 + j]) + (w1 - w2) * rth / Wt;
#pragma unroll
            for (int i = 0; i < nl; i++) acc += Kl()[(ns + i) * nz + j] * (-L->r0[S::R_LIN + i] * L->r1[S::R_LIN + i]);
#pragma unroll
            for (int r = 0; r < 4 * nsoc; r++) acc += Kl()[(ns + nl + r) * nz + j] * L->tmp[r];
            L->b[j] = acc;
        }
        if (np > 0) {
            for (int r = lane; r < nl + 4 * nsoc; r += 64) {
                const double tl = r < nl ? -L->r0[S::R_LIN + r] * L->r1[S::R_LIN + r] : L->tmp[r - nl];
#pragma unroll
                for (int j = 0; j < np; j++) bp[j] += Kp()[(ns + r) * npa + j] * tl;
            }
        }
        for (int c = lane; c < m; c += 64) {
            double t = 0.0;
            if (nu_live(k, c)) {
                double w1, w2, t1, t2, rxa; bool hg;
                nu_row_data(k, c, L->r0, L->r1, L->g0, L->g1, L->ak, L->ga, w1, w2, t1, t2, rxa, hg);
                const Pair pr = hg ? pairC(w1, w2, t1, t2, rxa) : pairA(w1, w2, t1, t2, rxa);
                t = pr.tau / pr.kap;
            }
            L->t[c] = t;
        }
        sync();
        // b-hat = Li b
        if (lane < nz) {
            double acc = 0.0;
            for (int q = 0; q <= lane; q++) acc += Li()[lane * nz + q] * L->b[q];
            L->bh[lane] = acc;
        }
        sync();
        // tp = t - Y' b-hat
        for (int c = lane; c < m; c += 64) {
            double acc = L->t[c];
#pragma unroll
            for (int q = 0; q < nz; q++) acc -= Ym()[q * MNU + c] * L->bh[q];
            L->thp[c] = acc;
        }
        sync();
        // t-hat = Lni tp
        for (int c = lane; c < m; c += 64) {
            double acc = 0.0;
            for (int q = 0; q <= c; q++) acc += Lni()[c * MNU + q] * L->thp[q];
            L->th[c] = acc;
        }
        sync();
        // X' t-hat for the next node; store b-hat, t-hat
        if (lane < nz) {
            double acc = 0.0;
            for (int r = 0; r < m; r++) acc += Xm()[r * nz + lane] * L->th[r];
            L->znx[lane] = acc;
            fb[(long)k * nz + lane] = L->bh[lane];
        }
        for (int c = lane; c < MNU; c += 64) ft[(long)k * MNU + c] = c < m ? L->th[c] : 0.0;
        sync();
    }
    prof[3] += tick() - t0s_;
    const long long tb_ = tick();
    // ---------------- backward sweep: nu = Lni'(X z+ - t-hat), z = Li'(b-hat - Y nu) ----------------
    prefetchF(N - 1);
    for (int k = N - 1; k >= 0; k--) {
        const int m = mnu(k);
        commitF();
        if (lane < nz) L->bh[lane] = fb[(long)k * nz + lane];
        for (int c = lane; c < m; c += 64) L->th[c] = ft[(long)k * MNU + c];
        sync();
        if (k > 0) prefetchF(k - 1);
        for (int c = lane; c < m; c += 64) {
            double acc = -L->th[c];
            if (k < N - 1) {
#pragma unroll
                for (int q = 0; q < nz; q++) acc += Xm()[c * nz + q] * L->zn[q];
            }
            L->thp[c] = acc;
        }
        sync();
        for (int c = lane; c < m; c += 64) {
            double acc = 0.0;
            for (int r = c; r < m; r++) acc += Lni()[r * MNU + c] * L->thp[r];
            L->nuk[c] = acc;
        }
        sync();
        if (lane < nz) {
            double acc = L->bh[lane];
            for (int c = 0; c < m; c++) acc -= Ym()[lane * MNU + c] * L->nuk[c];
            L->b[lane] = acc;
        }
        sync();
        if (lane < nz) {
            double acc = 0.0;
            for (int r = lane; r < nz; r++) acc += Li()[r * nz + lane] * L->b[r];
            L->zk[lane] = acc;
        }
        sync();
        if (lane < nz) { fb[(long)k * nz + lane] = L->zk[lane]; L->zn[lane] = L->zk[lane]; }
        for (int c = lane; c < MNU; c += 64) ft[(long)k * MNU + c] = c < m ? L->nuk[c] : 0.0;
        sync();
    }
    prof[4] += tick() - tb_;
    const long long t1s_ = tick();
    // ---------------- arrow: dp = Sp^-1 (bp - [C0; Ft]' y_b) ; z -= Ycz dp ; nu -= Ycnu dp ----------------
    double dp[npa];
#pragma unroll
    for (int j = 0; j < npa; j++) dp[j] = 0.0;
    if (np > 0) {
        const double* gC0 = W + wo.C0; const double* gYcz = W + wo.Ycz; const double* gYcnu = W + wo.Ycnu;
        prefetch(0);
        for (int k = 0; k < N; k++) {
            commit();
            sync();
            if (k + 1 < N) prefetch(k + 1);
            for (int r = lane; r < nz + MNU; r += 64) {
                double yv;
                if (r < nz) yv = fb[(long)k * nz + r];
                else { const int c = r - nz; if (c >= mnu(k) || !nu_live(k, c)) continue; yv = ft[(long)k * MNU + c]; }
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
        for (int i = 0; i < np; i++) { double v = dp[i]; for (int q = 0; q < i; q++) v -= spL[i * npa + q] * dp[q]; dp[i] = v / spL[i * npa + i]; }
        for (int i = np - 1; i >= 0; i--) { double v = dp[i]; for (int q = i + 1; q < np; q++) v -= spL[q * npa + i] * dp[q]; dp[i] = v / spL[i * npa + i]; }
        for (long idx = lane; idx < (long)N * nz; idx += 64) {
            double v = fb[idx];
#pragma unroll
            for (int j = 0; j < np; j++) v -= gYcz[idx * npa + j] * dp[j];
            fb[idx] = v;
        }
        for (long idx = lane; idx < (long)N * MNU; idx += 64) {
            double v = ft[idx];
#pragma unroll
            for (int j = 0; j < np; j++) v -= gYcnu[idx * npa + j] * dp[j];
            ft[idx] = v;
        }
        sync();
    }
    for (long idx = lane; idx < (long)N * nz; idx += 64) dxi[idx] = fb[idx];
    for (long idx = lane; idx < (long)N * MNU; idx += 64) nuv[idx] = ft[idx];
    if (lane < npa) PV(dxi, lane) = dp[lane];
    sync();
    prof[5] += tick() - t1s_;
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
    prefetch(0);
    for (int k = 0; k < N; k++) {
        const int m = mnu(k);
        commit();
        load_rows(L->r0, w, k); load_rows(L->r1, rtil, k);
        if (lane < nz) { L->zk[lane] = Z(dxi, k, lane); L->zn[lane] = (k < N - 1) ? Z(dxi, k + 1, lane) : 0.0; }
        if (lane < AS) L->ak[lane] = AUX(rxv, k, lane);
        for (int c = lane; c < MNU; c += 64) L->nuk[c] = nuv[(long)k * MNU + c];
        for (int i = lane; i < nsoc * 36; i += 64) L->soc[i] = socW[(long)k * nsoc * 36 + i];
        sync();
        if (k + 1 < N) prefetch(k + 1);
        for (int r = lane; r < RS; r += 64) L->arow[r] = row_main(k, r);
        // boundary-condition rows (global) handled at their node
        if (k == 0 || k == N - 1) {
            const int nb = k == 0 ? nic : ntc;
            const double* H = k == 0 ? gH0() : gHf();
            const double* K = k == 0 ? gK0() : gKf();
            if (k == 0 || N > 1) {
                for (int i = lane; i < nb; i += 64) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < nx; j++) acc += H[i * nx + j] * L->zk[j];
#pragma unroll
                    for (int j = 0; j < np; j++) acc += K[i * npa + j] * L->pv[j];
                    L->tmp[i] = acc;
                }
            }
        }
        sync();
        // ---- aux steps of this node ----
        if (lane < nx + ns) {
            const int c = lane;
            double val = 0.0;
            if (nu_live(k, c)) {
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
                if (k < N - 1) {
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
        if (k == 0 || k == N - 1) {
            const bool isic = k == 0;
            const int nb = isic ? nic : ntc;
            if (isic || N > 1) {
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
        }
        sync();
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
    sync();
    prof[6] += tick() - t0_;
}

}  // namespace scp
