// Main loop of the structured IPM v2 (included by ipm2_kernel.hpp).  Mirrors oracle/ipm_struct.py::solve.
#pragma once

namespace scp {

template <class M>
__device__ __forceinline__ void Ipm2<M>::nt_update(double* s, double* lam)
{
    double* socW = W + wo.socW;
    for (int idx = lane; idx < N * nsoc; idx += 64) {
        const int k = idx / nsoc, c = idx % nsoc;
        double sv[4], zv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { sv[q] = ROW(s, k, S::R_SOC + 4 * c + q); zv[q] = ROW(lam, k, S::R_SOC + 4 * c + q); }
        const double sres = sqrt(sv[0] * sv[0] - sv[1] * sv[1] - sv[2] * sv[2] - sv[3] * sv[3]);
        const double zres = sqrt(zv[0] * zv[0] - zv[1] * zv[1] - zv[2] * zv[2] - zv[3] * zv[3]);
        double sb[4], zb[4], dot = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) { sb[q] = sv[q] / sres; zb[q] = zv[q] / zres; dot += sb[q] * zb[q]; }
        const double gamma = sqrt((1.0 + dot) / 2.0);
        double wb[4];
        wb[0] = (sb[0] + zb[0]) / (2 * gamma);
#pragma unroll
        for (int q = 1; q < 4; q++) wb[q] = (sb[q] - zb[q]) / (2 * gamma);
        const double eta = sqrt(sres / zres);
        double* Wm = socW + (long)idx * 36;
        double* Wi = Wm + 16;
        double* lt = Wm + 32;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                double v;
                if (r == 0 && q == 0) v = wb[0];
                else if (r == 0) v = wb[q];
                else if (q == 0) v = wb[r];
                else v = (r == q ? 1.0 : 0.0) + wb[r] * wb[q] / (1.0 + wb[0]);
                Wm[r * 4 + q] = v * eta;
                Wi[r * 4 + q] = ((r == 0) != (q == 0) ? -v : v) / eta;
            }
#pragma unroll
        for (int r = 0; r < 4; r++) { double acc = 0.0; for (int q = 0; q < 4; q++) acc += Wm[r * 4 + q] * zv[q]; lt[r] = acc; }
        if (!(sres > 0.0) || !(zres > 0.0) || !isfinite(eta)) L->fail = 1;
    }
    gsync();
}

template <class M>
__device__ __forceinline__ void Ipm2<M>::nt_identity()
{
    double* socW = W + wo.socW;
    for (int idx = lane; idx < N * nsoc; idx += 64) {
        double* Wm = socW + (long)idx * 36;
        for (int q = 0; q < 16; q++) { Wm[q] = (q % 5 == 0) ? 1.0 : 0.0; Wm[16 + q] = Wm[q]; }
        for (int q = 0; q < 4; q++) Wm[32 + q] = 0.0;
    }
    gsync();
}

template <class M>
__device__ __forceinline__ double Ipm2<M>::soc_step(const double* s, const double* d)
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

template <class M>
__device__ __forceinline__ double Ipm2<M>::max_step(double* v, double* dv) const
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

template <class M>
__device__ __forceinline__ double Ipm2<M>::min_margin(double* v, double* dv, double alpha) const
{
    double mm = 1e300;
    for (int k = 0; k < N; k++)
        for (int r = lane; r < RS; r += 64) {
            if (!live(k, r)) continue;
            if (r < S::R_SOC) mm = fmin(mm, ROW(v, k, r) + (dv ? alpha * ROW(dv, k, r) : 0.0));
            else if ((r - S::R_SOC) % 4 == 0) {
                double t[4];
#pragma unroll
                for (int q = 0; q < 4; q++) t[q] = ROW(v, k, r + q) + (dv ? alpha * ROW(dv, k, r + q) : 0.0);
                mm = fmin(mm, t[0] - sqrt(t[1] * t[1] + t[2] * t[2] + t[3] * t[3]));
            }
        }
    for (int r = lane; r < RG; r += 64) mm = fmin(mm, GROW(v, r) + (dv ? alpha * GROW(dv, r) : 0.0));
    return wave_min(mm);
}

template <class M>
__device__ __forceinline__ void Ipm2<M>::run()
{
    static_assert(MNU * npa <= 64, "arrow scratch (tmp) too small for this model");
    const long XI = WK::XI(N), ROWS = WK::ROWS(N);
    double *xi = W + wo.xi, *dxi = W + wo.dxi, *rx = W + wo.rx, *exi = W + wo.exi, *best = W + wo.best, *rxe = W + wo.rxe,
           *cv = W + wo.cv, *qd = W + wo.qd;
    double *s = W + wo.s, *lam = W + wo.lam, *rz = W + wo.rz, *w = W + wo.w, *rtil = W + wo.rtil, *ds = W + wo.ds,
           *dl = W + wo.dl, *gd = W + wo.gd, *r2 = W + wo.r2, *el = W + wo.el, *hneg = W + wo.hneg, *ge = W + wo.ge;
    double* socW = W + wo.socW;
    const long long t_start_ = tick();
    if (lane == 0) L->fail = 0;
    for (int i = lane; i < GR; i += 64) L->G[i] = Pg[o.glob + i];
    gsync();
    build_constants(hneg, cv, qd);

    double nh = 0.0, nc = 0.0, deg = 0.0;
    for (long i = lane; i < ROWS; i += 64) nh += hneg[i] * hneg[i];
    for (long i = lane; i < XI; i += 64) nc += cv[i] * cv[i];
    const double nrm_h = fmax(1.0, sqrt(wave_sum(nh))), nrm_c = fmax(1.0, sqrt(wave_sum(nc)));
    for (int k = 0; k < N; k++)
        for (int r = lane; r < S::R_SOC; r += 64) if (live(k, r)) deg += 1.0;
    for (int r = lane; r < RG; r += 64) deg += 1.0;
    deg = wave_sum(deg) + (double)N * nsoc;

    // One loop drives the initial point (it == -1: weights 1, cones W = I, r~z = -h, rx = c) and the
    // Mehrotra iterations, so that factor / newton_solve / finish_direction have a single (inlined) call site.
    int status = IPM_ITERLIM;
    int it = 0, best_it = 0;
    double best_merit = 1e300;
    double info_best[7] = {0, 0, 0, 0, 0, 0, 1e300};
    double gap = 0.0, mu = 0.0, sigma = 0.0, relgap_it = 1e300;
    for (it = -1; it <= a.max_iter; it++) {
        if (it < 0) {
            for (long i = lane; i < ROWS; i += 64) { w[i] = 1.0; rtil[i] = hneg[i]; }
            for (long i = lane; i < XI; i += 64) { rx[i] = cv[i]; xi[i] = 0.0; dxi[i] = 0.0; }
            nt_identity();
            gsync();
        } else {
            // ---- residuals ----
            GT_apply(lam, rx);
            G_apply(xi, gd);
            double lrz = 0.0, nrz = 0.0, nrx = 0.0, pc = 0.0;
            gap = 0.0;
            for (long i = lane; i < XI; i += 64) {
                const double x_ = xi[i], q_ = qd[i], c_ = cv[i];
                const double r_ = rx[i] + q_ * x_ + c_;
                rx[i] = r_;
                nrx += r_ * r_;
                pc += 0.5 * q_ * x_ * x_ + c_ * x_;
            }
            for (int k = 0; k < N; k++)
                for (int r = lane; r < RS; r += 64) {
                    const bool lv = live(k, r);
                    const double v = lv ? ROW(gd, k, r) + ROW(s, k, r) + ROW(hneg, k, r) : 0.0;
                    ROW(rz, k, r) = v;
                    if (lv) { gap += ROW(s, k, r) * ROW(lam, k, r); lrz += ROW(lam, k, r) * v; nrz += v * v; }
                }
            for (int r = lane; r < RG; r += 64) {
                const double v = GROW(gd, r) + GROW(s, r) + GROW(hneg, r);
                GROW(rz, r) = v;
                gap += GROW(s, r) * GROW(lam, r); lrz += GROW(lam, r) * v; nrz += v * v;
            }
            gsync();
            gap = wave_sum(gap); lrz = wave_sum(lrz); nrz = wave_sum(nrz); nrx = wave_sum(nrx);
            const double pcost = wave_sum(pc);
            const double dcost = pcost + lrz - gap;
            const double pres = sqrt(nrz) / nrm_h, dres = sqrt(nrx) / nrm_c;
            const double relgap = pcost < 0.0 ? gap / -pcost : (dcost > 0.0 ? gap / dcost : 1e300);
            relgap_it = relgap;
            const double merit = fmax(fmax(pres / a.feastol, dres / a.feastol), fmin(gap / a.abstol, relgap / a.reltol));
            const bool finite_ok = isfinite(merit) && (L->fail == 0);
            if (finite_ok && merit < best_merit) {
                best_merit = merit; best_it = it;
                for (long i = lane; i < XI; i += 64) best[i] = xi[i];
                info_best[0] = pcost + cost_const; info_best[1] = dcost + cost_const; info_best[2] = gap; info_best[3] = pres;
                info_best[4] = dres; info_best[5] = relgap; info_best[6] = merit;
                gsync();
            }
            if (!finite_ok) { status = IPM_NUMERR; break; }
            if (merit <= 1.0) { status = IPM_OPTIMAL; break; }
            if (it == a.max_iter) break;
            if (best_merit <= 1e3 && it - best_it >= a.stall) break;
            // ---- scalings ----
            for (int k = 0; k < N; k++)
                for (int r = lane; r < S::R_SOC; r += 64) ROW(w, k, r) = live(k, r) ? ROW(lam, k, r) / ROW(s, k, r) : 1.0;
            for (int r = lane; r < RG; r += 64) GROW(w, r) = GROW(lam, r) / GROW(s, r);
            gsync();
            nt_update(s, lam);
            if (L->fail) { status = IPM_NUMERR; break; }
            mu = gap / deg;
        }
        factor(w);
        if (L->fail) { status = IPM_NUMERR; break; }
        const int nphase = it < 0 ? 1 : 2;
        for (int phase = 0; phase < nphase; phase++) {
            if (it >= 0 && phase == 0) {
                // affine direction: r~z = rz - s
                for (long i = lane; i < ROWS; i += 64) rtil[i] = rz[i] - s[i];
                gsync();
            } else if (it >= 0) {
                // combined direction: r~z = rz - s + (sigma mu - ds_a dl_a)/lam ; cones: rz + W (lam~ \ d_s)
                for (int k = 0; k < N; k++)
                    for (int r = lane; r < RS; r += 64) {
                        double v = ROW(rz, k, r) - ROW(s, k, r);
                        if (live(k, r)) {
                            if (r < S::R_SOC) v += (sigma * mu - ROW(ds, k, r) * ROW(dl, k, r)) / ROW(lam, k, r);
                            else if ((r - S::R_SOC) % 4 == 0) {
                                const int c = (r - S::R_SOC) / 4;
                                const double* Wm = socW + ((long)k * nsoc + c) * 36;
                                const double* Wi = Wm + 16;
                                const double* lt = Wm + 32;
                                double u1[4], u2[4], dsv[4];
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    double a1 = 0.0, a2 = 0.0;
#pragma unroll
                                    for (int q2 = 0; q2 < 4; q2++) { a1 += Wi[q * 4 + q2] * ROW(ds, k, r + q2); a2 += Wm[q * 4 + q2] * ROW(dl, k, r + q2); }
                                    u1[q] = a1; u2[q] = a2;
                                }
                                dsv[0] = sigma * mu - (lt[0] * lt[0] + lt[1] * lt[1] + lt[2] * lt[2] + lt[3] * lt[3]) -
                                         (u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2] + u1[3] * u2[3]);
#pragma unroll
                                for (int q = 1; q < 4; q++) dsv[q] = -2.0 * lt[0] * lt[q] - (u1[0] * u2[q] + u2[0] * u1[q]);
                                const double den = lt[0] * lt[0] - lt[1] * lt[1] - lt[2] * lt[2] - lt[3] * lt[3];
                                double uu[4];
                                uu[0] = (lt[0] * dsv[0] - lt[1] * dsv[1] - lt[2] * dsv[2] - lt[3] * dsv[3]) / den;
#pragma unroll
                                for (int q = 1; q < 4; q++) uu[q] = (dsv[q] - uu[0] * lt[q]) / lt[0];
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    double acc = 0.0;
#pragma unroll
                                    for (int q2 = 0; q2 < 4; q2++) acc += Wm[q * 4 + q2] * uu[q2];
                                    ROW(rtil, k, r + q) = ROW(rz, k, r + q) + acc;
                                }
                                continue;
                            } else continue;
                        }
                        ROW(rtil, k, r) = v;
                    }
                for (int r = lane; r < RG; r += 64)
                    GROW(rtil, r) = GROW(rz, r) - GROW(s, r) + (sigma * mu - GROW(ds, r) * GROW(dl, r)) / GROW(lam, r);
                gsync();
            }
            // ---- Newton solve + iterative refinement in augmented form ----
            // refinement only once the gap is small (the Newton system is well conditioned early on)
            const int nref_eff = (it < 0 || !(relgap_it < a.ref_gap)) ? 0 : a.nref;
            for (int rf = 0; rf <= nref_eff; rf++) {
                double *rt_ = rtil, *rx_ = rx, *ox = (it < 0 ? xi : dxi), *og = gd, *ol = dl;
                if (rf > 0) {
                    // -r1 = rx + P dxi + G'dl   (rxe) ;  -r2 = r~z + gd - W^2 dl   (r2)
                    GT_apply(dl, rxe);
                    for (long i = lane; i < XI; i += 64) rxe[i] += qd[i] * dxi[i] + rx[i];
                    for (int k = 0; k < N; k++)
                        for (int r = lane; r < RS; r += 64) {
                            double v = 0.0;
                            if (live(k, r)) {
                                if (r < S::R_SOC) v = ROW(rtil, k, r) + ROW(gd, k, r) - ROW(dl, k, r) / ROW(w, k, r);
                                else {
                                    const int c = (r - S::R_SOC) / 4, rr = (r - S::R_SOC) % 4;
                                    const double* Wm = socW + ((long)k * nsoc + c) * 36;
                                    double t1[4];
#pragma unroll
                                    for (int q = 0; q < 4; q++) { double acc = 0.0; for (int q2 = 0; q2 < 4; q2++) acc += Wm[q * 4 + q2] * ROW(dl, k, S::R_SOC + 4 * c + q2); t1[q] = acc; }
                                    double acc = 0.0;
#pragma unroll
                                    for (int q = 0; q < 4; q++) acc += Wm[rr * 4 + q] * t1[q];
                                    v = ROW(rtil, k, r) + ROW(gd, k, r) - acc;
                                }
                            }
                            ROW(r2, k, r) = v;
                        }
                    for (int r = lane; r < RG; r += 64) GROW(r2, r) = GROW(rtil, r) + GROW(gd, r) - GROW(dl, r) / GROW(w, r);
                    gsync();
                    rt_ = r2; rx_ = rxe; ox = exi; og = ge; ol = el;
                }
                newton_solve(w, rt_, rx_, ox);
                finish_direction(w, rt_, rx_, ox, og, ol);
                if (rf > 0) {
                    for (long i = lane; i < XI; i += 64) dxi[i] += exi[i];
                    for (long i = lane; i < ROWS; i += 64) { dl[i] += el[i]; gd[i] += ge[i]; }
                    gsync();
                }
            }
            if (it < 0) {
                // initial point: lam = G xi - h ; s = -lam ; shift into the cone
                for (long i = lane; i < ROWS; i += 64) { lam[i] = gd[i] + hneg[i]; s[i] = -lam[i]; }
                gsync();
                for (int r = lane; r < 2 * nx; r += 64) { ROW(lam, N - 1, r) = 1.0; ROW(s, N - 1, r) = 1.0; }  // dead rows
                gsync();
                for (int pass = 0; pass < 2; pass++) {
                    double* v = pass == 0 ? s : lam;
                    const double mm = min_margin(v, nullptr, 0.0);
                    if (mm <= 0.0) {
                        const double sh = 1.0 - mm;
                        for (int k = 0; k < N; k++)
                            for (int r = lane; r < RS; r += 64) {
                                if (!live(k, r)) continue;
                                if (r < S::R_SOC || (r - S::R_SOC) % 4 == 0) ROW(v, k, r) += sh;
                            }
                        for (int r = lane; r < RG; r += 64) GROW(v, r) += sh;
                    }
                    gsync();
                }
            } else {
                for (long i = lane; i < ROWS; i += 64) ds[i] = -rz[i] - gd[i];
                gsync();
                if (phase == 0) {
                    const double a_aff = fmin(1.0, fmin(max_step(s, ds), max_step(lam, dl)));
                    sigma = (1.0 - a_aff) * (1.0 - a_aff) * (1.0 - a_aff);
                } else {
                    double alpha = fmin(1.0, 0.99 * fmin(max_step(s, ds), max_step(lam, dl)));
                    for (int bt = 0; bt < 60; bt++) {
                        if (min_margin(s, ds, alpha) > 0.0 && min_margin(lam, dl, alpha) > 0.0) break;
                        alpha *= 0.8;
                    }
                    for (long i = lane; i < XI; i += 64) xi[i] += alpha * dxi[i];
                    for (int k = 0; k < N; k++)
                        for (int r = lane; r < RS; r += 64)
                            if (live(k, r)) { ROW(s, k, r) += alpha * ROW(ds, k, r); ROW(lam, k, r) += alpha * ROW(dl, k, r); }
                    for (int r = lane; r < RG; r += 64) { GROW(s, r) += alpha * GROW(ds, r); GROW(lam, r) += alpha * GROW(dl, r); }
                    gsync();
                }
            }
        }
    }
    if (it < 0) it = 0;
    prof[7] = tick() - t_start_;
    // ---------------- result: best iterate ----------------
    if (status != IPM_OPTIMAL) {
        // ECOS "reduced tolerances" -> ALMOST_OPTIMAL
        if (info_best[3] <= 1e-4 && info_best[4] <= 1e-4 && (info_best[2] <= 5e-5 || info_best[5] <= 5e-5)) status = IPM_ALMOST;
    }
    for (long i = lane; i < (long)N * nz; i += 64) a.z_out[(long)blockIdx.x * N * nz + i] = best[i];
    if (lane < npa) a.p_out[(long)blockIdx.x * npa + lane] = PV(best, lane);
    if (lane == 0) {
        a.status[blockIdx.x] = status;
        a.iters[blockIdx.x] = it;
        for (int i = 0; i < 7; i++) a.info[(long)blockIdx.x * 8 + i] = info_best[i];
        a.info[(long)blockIdx.x * 8 + 7] = (double)best_it;
        if (a.prof) for (int i = 0; i < 8; i++) a.prof[(long)blockIdx.x * 8 + i] = prof[i];
    }
}

template <class M>
__global__ __launch_bounds__(64) void ipm2_solve_kernel(IpmArgs a)
{
    if (a.active != nullptr && a.active[blockIdx.x] == 0) return;
    __shared__ typename Ipm2<M>::Lds lds;
    Ipm2<M> S_;
    S_.a = a;
    S_.N = a.N;
    S_.lane = threadIdx.x;
    S_.Pg = a.slab + (long)blockIdx.x * a.slab_stride;
    S_.o = SP<M>::offsets(a.N);
    S_.wo = Ipm2Work<M>::offsets(a.N);
    S_.W = a.work + (long)blockIdx.x * a.work_stride;
    S_.L = &lds;
    S_.ttrp = S_.Pg[S_.o.scal + 0];
    S_.cost_const = S_.Pg[S_.o.scal + 1];
    S_.run();
}

}  // namespace scp
