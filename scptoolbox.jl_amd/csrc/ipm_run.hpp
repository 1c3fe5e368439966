// Main loop of the structured IPM (included by ipm_kernel.hpp).  Mirrors oracle/ipm_struct.py::solve.
#pragma once

namespace scp {

template <class M>
__device__ void Ipm<M>::run()
{
    const long XI = WK::XI(N), ROWS = WK::ROWS(N);
    double *xi = W + wo.xi, *dxi = W + wo.dxi, *rx = W + wo.rx, *exi = W + wo.exi, *best = W + wo.best, *rxe = W + wo.rxe;
    double *s = W + wo.s, *lam = W + wo.lam, *rz = W + wo.rz, *w = W + wo.w, *rtil = W + wo.rtil, *ds = W + wo.ds,
           *dl = W + wo.dl, *gd = W + wo.gd, *r2 = W + wo.r2, *el = W + wo.el, *hneg = W + wo.hneg, *ge = W + wo.ge;
    double* socW = W + wo.socW;

    if (lane == 0) L->fail = 0;
    const long long t_start_ = tick();
    build_hneg(hneg);

    // cost vector on the xi layout, written into a helper lambda
    auto add_cost = [&](double* v, double scale_existing) {
        for (int k = 0; k < N; k++) {
            if (lane < nz) Z(v, k, lane) = scale_existing * Z(v, k, lane) + cvec(0, k, lane);
            else if (lane < nz + AS) AUX(v, k, lane - nz) = scale_existing * AUX(v, k, lane - nz) + cvec(1, k, lane - nz);
        }
        if (lane < npa) PV(v, lane) = scale_existing * PV(v, lane) + (np > 0 ? cvec(2, 0, lane) : 0.0);
        for (int i = lane; i < AG; i += 64) GAUX(v, i) = scale_existing * GAUX(v, i) + cvec(3, 0, i);
        sync();
    };

    // norms of h and c (termination scaling, as in oracle/ipm.py)
    double nh = 0.0, nc = 0.0;
    for (long i = lane; i < ROWS; i += 64) nh += hneg[i] * hneg[i];
    for (long i = lane; i < XI; i += 64) rx[i] = 0.0;
    sync();
    add_cost(rx, 0.0);
    for (long i = lane; i < XI; i += 64) nc += rx[i] * rx[i];
    const double nrm_h = fmax(1.0, sqrt(wave_sum(nh))), nrm_c = fmax(1.0, sqrt(wave_sum(nc)));
    double deg = 0.0;
    for (int k = 0; k < N; k++)
        for (int r = lane; r < S::R_SOC; r += 64) if (live(k, r)) deg += 1.0;
    for (int r = lane; r < RG; r += 64) deg += 1.0;
    deg = wave_sum(deg) + (double)N * nsoc;

    // ---------------- initial point: (P + G'G) xi = -c + G'h ; lam = G xi - h ; s = -lam ; shift ----------------
    for (long i = lane; i < ROWS; i += 64) { w[i] = 1.0; rtil[i] = hneg[i]; }
    nt_identity();
    sync();
    factor(w);
    newton_solve(w, rtil, rx, xi);
    G_apply(xi, gd);
    for (long i = lane; i < ROWS; i += 64) { lam[i] = gd[i] + hneg[i]; s[i] = -lam[i]; }
    sync();
    for (int r = lane; r < 2 * nx; r += 64) { ROW(lam, N - 1, r) = 1.0; ROW(s, N - 1, r) = 1.0; }  // dead rows
    sync();
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
        sync();
    }

    int status = IPM_ITERLIM;
    int it = 0, best_it = 0;
    double best_merit = 1e300;
    double info_best[7] = {0, 0, 0, 0, 0, 0, 1e300};
    for (it = 0; it <= a.max_iter; it++) {
        // ---- residuals ----
        GT_apply(lam, rx);
        for (int k = 0; k < N; k++)
            if (lane < nz) Z(rx, k, lane) += P[o.Qd(k) + lane] * Z(xi, k, lane);
        if (lane < np) PV(rx, lane) += P[o.Qp + lane] * PV(xi, lane);
        sync();
        add_cost(rx, 1.0);
        G_apply(xi, gd);
        double gap = 0.0, lrz = 0.0, nrz = 0.0, nrx = 0.0, pc = 0.0;
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
        for (long i = lane; i < XI; i += 64) nrx += rx[i] * rx[i];
        for (int k = 0; k < N; k++) {
            if (lane < nz) { const double zz = Z(xi, k, lane); pc += 0.5 * P[o.Qd(k) + lane] * zz * zz + cvec(0, k, lane) * zz; }
            else if (lane < nz + AS) pc += cvec(1, k, lane - nz) * AUX(xi, k, lane - nz);
        }
        if (lane < np) { const double pv = PV(xi, lane); pc += 0.5 * P[o.Qp + lane] * pv * pv + cvec(2, 0, lane) * pv; }
        for (int i = lane; i < AG; i += 64) pc += cvec(3, 0, i) * GAUX(xi, i);
        sync();
        gap = wave_sum(gap); lrz = wave_sum(lrz); nrz = wave_sum(nrz); nrx = wave_sum(nrx);
        const double pcost = wave_sum(pc);
        const double dcost = pcost + lrz - gap;
        const double pres = sqrt(nrz) / nrm_h, dres = sqrt(nrx) / nrm_c;
        const double relgap = pcost < 0.0 ? gap / -pcost : (dcost > 0.0 ? gap / dcost : 1e300);
        const double merit = fmax(fmax(pres / a.feastol, dres / a.feastol), fmin(gap / a.abstol, relgap / a.reltol));
        const bool finite_ok = isfinite(merit) && (L->fail == 0);
        if (finite_ok && merit < best_merit) {
            best_merit = merit; best_it = it;
            for (long i = lane; i < XI; i += 64) best[i] = xi[i];
            info_best[0] = pcost + cost_const; info_best[1] = dcost + cost_const; info_best[2] = gap; info_best[3] = pres;
            info_best[4] = dres; info_best[5] = relgap; info_best[6] = merit;
            sync();
        }
        if (!finite_ok) { status = IPM_NUMERR; break; }
        if (merit <= 1.0) { status = IPM_OPTIMAL; break; }
        if (it == a.max_iter) break;
        if (best_merit <= 1e3 && it - best_it >= a.stall) break;

        // ---- scalings + factorisation ----
        for (int k = 0; k < N; k++)
            for (int r = lane; r < S::R_SOC; r += 64) ROW(w, k, r) = live(k, r) ? ROW(lam, k, r) / ROW(s, k, r) : 1.0;
        for (int r = lane; r < RG; r += 64) GROW(w, r) = GROW(lam, r) / GROW(s, r);
        sync();
        nt_update(s, lam);
        if (L->fail) { status = IPM_NUMERR; break; }
        factor(w);
        if (L->fail) { status = IPM_NUMERR; break; }
        const double mu = gap / deg;

        auto newton_refined = [&](double* rt) {
            newton_solve(w, rt, rx, dxi);
            G_apply(dxi, gd);
            dlam_from(w, gd, rt, rx, dl);
            for (int rf = 0; rf < a.nref; rf++) {
                // -r1 = rx + P dxi + G'dl   (rxe) ;  -r2 = rt + gd - W^2 dl   (r2)
                GT_apply(dl, rxe);
                for (int k = 0; k < N; k++)
                    if (lane < nz) Z(rxe, k, lane) += P[o.Qd(k) + lane] * Z(dxi, k, lane);
                if (lane < np) PV(rxe, lane) += P[o.Qp + lane] * PV(dxi, lane);
                sync();
                for (long i = lane; i < XI; i += 64) rxe[i] += rx[i];
                for (int k = 0; k < N; k++)
                    for (int r = lane; r < RS; r += 64) {
                        double v = 0.0;
                        if (live(k, r)) {
                            if (r < S::R_SOC) v = ROW(rt, k, r) + ROW(gd, k, r) - ROW(dl, k, r) / ROW(w, k, r);
                            else {
                                const int c = (r - S::R_SOC) / 4, rr = (r - S::R_SOC) % 4;
                                const double* Wm = socW + ((long)k * nsoc + c) * 36;
                                double t1[4];
                                for (int q = 0; q < 4; q++) { double acc = 0.0; for (int q2 = 0; q2 < 4; q2++) acc += Wm[q * 4 + q2] * ROW(dl, k, S::R_SOC + 4 * c + q2); t1[q] = acc; }
                                double acc = 0.0;
                                for (int q = 0; q < 4; q++) acc += Wm[rr * 4 + q] * t1[q];
                                v = ROW(rt, k, r) + ROW(gd, k, r) - acc;
                            }
                        }
                        ROW(r2, k, r) = v;
                    }
                for (int r = lane; r < RG; r += 64) GROW(r2, r) = GROW(rt, r) + GROW(gd, r) - GROW(dl, r) / GROW(w, r);
                sync();
                newton_solve(w, r2, rxe, exi);
                G_apply(exi, ge);
                dlam_from(w, ge, r2, rxe, el);
                for (long i = lane; i < XI; i += 64) dxi[i] += exi[i];
                for (long i = lane; i < ROWS; i += 64) { dl[i] += el[i]; gd[i] += ge[i]; }
                sync();
            }
        };

        // ---- affine direction: r~z = rz - s ----
        for (long i = lane; i < ROWS; i += 64) rtil[i] = rz[i] - s[i];
        sync();
        newton_refined(rtil);
        for (long i = lane; i < ROWS; i += 64) ds[i] = -rz[i] - gd[i];
        sync();
        const double a_aff = fmin(1.0, fmin(max_step(s, ds), max_step(lam, dl)));
        const double sigma = (1.0 - a_aff) * (1.0 - a_aff) * (1.0 - a_aff);
        // ---- combined direction ----
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
                        for (int q = 0; q < 4; q++) {
                            double a1 = 0.0, a2 = 0.0;
                            for (int q2 = 0; q2 < 4; q2++) { a1 += Wi[q * 4 + q2] * ROW(ds, k, r + q2); a2 += Wm[q * 4 + q2] * ROW(dl, k, r + q2); }
                            u1[q] = a1; u2[q] = a2;
                        }
                        // d_s = sigma mu e - lt o lt - u1 o u2
                        dsv[0] = sigma * mu - (lt[0] * lt[0] + lt[1] * lt[1] + lt[2] * lt[2] + lt[3] * lt[3]) -
                                 (u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2] + u1[3] * u2[3]);
                        for (int q = 1; q < 4; q++) dsv[q] = -2.0 * lt[0] * lt[q] - (u1[0] * u2[q] + u2[0] * u1[q]);
                        // u = lt \ d_s  (Jordan inverse), then r~z = rz + W u
                        const double den = lt[0] * lt[0] - lt[1] * lt[1] - lt[2] * lt[2] - lt[3] * lt[3];
                        double uu[4];
                        uu[0] = (lt[0] * dsv[0] - lt[1] * dsv[1] - lt[2] * dsv[2] - lt[3] * dsv[3]) / den;
                        for (int q = 1; q < 4; q++) uu[q] = (dsv[q] - uu[0] * lt[q]) / lt[0];
                        for (int q = 0; q < 4; q++) {
                            double acc = 0.0;
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
        sync();
        newton_refined(rtil);
        for (long i = lane; i < ROWS; i += 64) ds[i] = -rz[i] - gd[i];
        sync();
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
        sync();
    }
    prof[7] = tick() - t_start_;
    // ---------------- result: best iterate ----------------
    if (status != IPM_OPTIMAL) {
        // ECOS "reduced tolerances" -> ALMOST_OPTIMAL
        if (info_best[3] <= 1e-4 && info_best[4] <= 1e-4 && (info_best[2] <= 5e-5 || info_best[5] <= 5e-5)) status = IPM_ALMOST;
    }
    for (int k = 0; k < N; k++)
        if (lane < nz) a.z_out[((long)blockIdx.x * N + k) * nz + lane] = Z(best, k, lane);
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
__global__ __launch_bounds__(64) void ipm_solve_kernel(IpmArgs a)
{
    if (a.active != nullptr && a.active[blockIdx.x] == 0) return;
    __shared__ typename Ipm<M>::Lds lds;
    Ipm<M> S_;
    S_.a = a;
    S_.N = a.N;
    S_.lane = threadIdx.x;
    S_.P = a.slab + (long)blockIdx.x * a.slab_stride;
    S_.o = SP<M>::offsets(a.N);
    S_.wo = IpmWork<M>::offsets(a.N);
    S_.W = a.work + (long)blockIdx.x * a.work_stride;
    S_.L = &lds;
    S_.ttrp = S_.P[S_.o.scal + 0];
    S_.cost_const = S_.P[S_.o.scal + 1];
    S_.run();
}

}  // namespace scp
