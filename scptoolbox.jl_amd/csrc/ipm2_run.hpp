// Main loop of the structured IPM (included by ipm2_kernel.hpp).  Mirrors oracle/ipm_struct.py::solve.
#pragma once

namespace scp {

template <class M>
__device__ __forceinline__ void Ipm2<M>::nt_update(double* s, double* lam)
{
    double* socW = W + wo.socW;
    for (int idx = lane; idx < N * nsoc; idx += 64) {
        const int k = idx / NSOC1, c = idx % NSOC1;
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

// ------------------------------------------------------------------------------------------------
// Phase functions.  The sweeps over the horizon are compiled as SEPARATE (non-inlined) device functions: inlined
// into one kernel body the register allocator sees ~30 loop nests at once, keeps dozens of loop-invariant
// values of every phase alive across all of them and ends up spilling to scratch INSIDE the hot loops (a
// scratch reload is a vmcnt(0) wait, i.e. it also drains the software prefetch).  As functions, each phase gets
// its own allocation and the only state crossing a call is what run() really carries.  The Ipm2 object is
// rebuilt from five uniform scalars; LDS is the same function-scope static in every phase (ipm2_lds).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// uniform pointer into GLOBAL memory: function arguments arrive in VGPRs as generic pointers; readfirstlane makes
// them scalar again and the round trip through address space 1 lets the compiler emit global_load/global_store
// (a flat access would tie up the LDS counter as well)
template <class T>
__device__ __forceinline__ T* uni(T* p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    typedef __attribute__((address_space(1))) T* gptr;
    return (T*)(gptr)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double uni(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// one LDS object per (model, kernel variant): an LDS variable reachable from a single kernel is addressed with
// absolute offsets; shared between kernels it is reached through a per-kernel lookup table (measured: -13 %)
template <class M, int WPE>
__device__ __forceinline__ typename Ipm2<M>::Lds* ipm2_lds()
{
    __shared__ typename Ipm2<M>::Lds lds;
    return &lds;
}
template <class M, int WPE>
__device__ __forceinline__ void ipm2_bind(Ipm2<M>& S, const double* Pg, double* W, int N, double reg)
{
    S.N = uni(N);
    S.lane = threadIdx.x;
    S.Pg = uni(Pg);
    S.W = uni(W);
    S.L = ipm2_lds<M, WPE>();
    S.o = SP<M>::offsets(S.N);
    S.wo = Ipm2Work<M>::offsets(S.N);
    S.a.reg = uni(reg);
}
#define SCP_PHASE __device__ __attribute__((noinline))
// phase clocks taken by the CALLER (-DSCP_IPM_PROF_CALLER, `make profc`): slots G 0, G' 1, factor 2, newton 3, nt 4, finish 6 include the
// call itself (argument set-up, the callee-saved registers a phase function spills and reloads); the in-function clocks are off
#ifdef SCP_IPM_PROF_CALLER
#define PH_T(i, call) do { const long long t_ = (long long)wall_clock64(); call; if (lane == 0) L->prof[i] += (long long)wall_clock64() - t_; } while (0)
#else
#define PH_T(i, call) do { call; } while (0)
#endif
template <class M, int WPE>
SCP_PHASE void ipm2_ph_G(const double* Pg, double* W, int N, double* v, double* out)
{
    Ipm2<M> S; ipm2_bind<M, WPE>(S, Pg, W, N, 0.0); S.G_apply(uni(v), uni(out));
}
template <class M, int WPE>
SCP_PHASE void ipm2_ph_GT(const double* Pg, double* W, int N, double* mu, double* out)
{
    Ipm2<M> S; ipm2_bind<M, WPE>(S, Pg, W, N, 0.0); S.GT_apply(uni(mu), uni(out));
}
template <class M, int WPE>
SCP_PHASE void ipm2_ph_factor(const double* Pg, double* W, int N, double reg, double* w)
{
    Ipm2<M> S; ipm2_bind<M, WPE>(S, Pg, W, N, reg); S.factor(uni(w));
}
template <class M, int WPE>
SCP_PHASE void ipm2_ph_newton(const double* Pg, double* W, int N, double* w, double* rtil, double* rxv, double* dxi)
{
    Ipm2<M> S; ipm2_bind<M, WPE>(S, Pg, W, N, 0.0); S.newton_solve(uni(w), uni(rtil), uni(rxv), uni(dxi));
}
template <class M, int WPE>
SCP_PHASE void ipm2_ph_finish(const double* Pg, double* W, int N, double* w, double* rtil, double* rxv, double* dxi, double* gd, double* dl)
{
    Ipm2<M> S; ipm2_bind<M, WPE>(S, Pg, W, N, 0.0); S.finish_direction(uni(w), uni(rtil), uni(rxv), uni(dxi), uni(gd), uni(dl));
}
template <class M, int WPE>
SCP_PHASE void ipm2_ph_nt(const double* Pg, double* W, int N, double* s, double* lam)
{
    Ipm2<M> S; ipm2_bind<M, WPE>(S, Pg, W, N, 0.0); S.nt_update(uni(s), uni(lam));
}

template <class M>
template <int WPE>
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
    if (lane == 0) { L->fail = 0; for (int i = 0; i < 8; i++) L->prof[i] = 0; }
    for (int i = lane; i < GR; i += 64) L->G[i] = Pg[o.glob + i];
    gsync();
    build_constants(hneg, cv, qd);

    // Row-vector sweeps below are FLAT over the [N][RS] + [RG] layout and batched (flat<>): U*NA loads are in
    // flight per lane before the first store, which is what a single wave needs to stream at more than one
    // element per memory round trip.  Rows of the last node's (absent) dynamics are "dead": s = lam = 1, all
    // directions 0, excluded from the reductions.
    const int dead0 = (N - 1) * RS, dead1 = dead0 + 2 * nx, nrow_stage = N * RS;
    auto is_dead = [&](int i) { return i >= dead0 && i < dead1; };
    auto is_soc = [&](int i) { return nsoc > 0 && i < nrow_stage && (i % RS) >= S::R_SOC; };
    const int ncone = N * nsoc;
    auto cone_base = [&](int idx) { return (idx / NSOC1) * RS + S::R_SOC + 4 * (idx % NSOC1); };

    double nh = 0.0, nc = 0.0, deg = 0.0;
    {
        const double* in1[1] = {hneg};
        flat<1, 8>(ROWS, in1, [&](long, const double(&v)[1]) { nh += v[0] * v[0]; });
        const double* in2[1] = {cv};
        flat<1, 8>(XI, in2, [&](long, const double(&v)[1]) { nc += v[0] * v[0]; });
    }
    const double nrm_h = fmax(1.0, sqrt(wave_sum(nh))), nrm_c = fmax(1.0, sqrt(wave_sum(nc)));
    double nq = 0.0;
    {
        const double* in3[1] = {qd};
        flat<1, 8>(XI, in3, [&](long, const double(&v)[1]) { nq += v[0] * v[0]; });
    }
    const bool has_quad = wave_sum(nq) > 0.0;   // quadratic cost term present (P != 0)
    for (int i = lane; i < (int)ROWS; i += 64) if (!is_dead(i) && !is_soc(i)) deg += 1.0;
    deg = wave_sum(deg) + (double)N * nsoc;

    // One loop drives the initial point (it == -1: weights 1, cones W = I, r~z = -h, rx = c) and the
    // Mehrotra iterations, so that factor / newton_solve / finish_direction have a single (inlined) call site.
    // Warm start (scp_ptr_params.ipm_warm): attempt 0 starts from a snapshot of the previous launch (below); if that solve fails,
    // attempt 1 repeats it cold.  Same algebra as oracle/cpu_ptr.cpp (CpuIpm::solve).
    int status = IPM_ITERLIM;
    int it = 0, best_it = 0, iters_total = 0;
    double best_merit = 1e300;
    double info_best[7] = {0, 0, 0, 0, 0, 0, 1e300};
    double gap = 0.0, mu = 0.0, sigma = 0.0, relgap_it = 1e300;
    // Warm start from a SNAPSHOT of the previous solve of this problem (round 4; before: the previous FINAL iterate pushed back
    // into the interior, which left the products s_i lam_i spread over ten decades and needed ~31 iterations from mu = 1e-5).
    // A snapshot is an iterate of the previous solve itself, i.e. a point close to ITS central path; the data of two successive
    // PTR subproblems differ by the reference deviation, so for a small deviation the fine snapshot (mu ~ 1e-7) is nearly centred
    // for the new problem too (18 iterations instead of 32 on the rocket batch), and for a large one the coarse snapshot
    // (mu ~ 1e-1) still saves the ~10 iterations the cold start spends coming down from mu ~ 1e4 (PTR iterations 2-7: 25-37
    // instead of 31-57).  The infeasible-start iteration absorbs the change of the data.  Same algebra in oracle/cpu_ptr.cpp.
    // Round 6: four levels, chosen by the SCALE of the change -- the finest level l with prev_dev <= warm_dev[l] whose snapshot exists.
    // The new problem's residual at an old iterate is of the order of the reference deviation; an iterate much closer to the boundary
    // than that (mu = 1e-10 after a move of 1e-4) crawls with steps of 0.01 until the warm attempt is abandoned, a matching one needs
    // 1 ... 5 iterations (oracle/cpu_ptr.cpp: same rule, levels swept there first).
    // (the coarse level only pays where cold solves are slow: it is used when the last cold solve of the problem needed at
    // least warm_min_cold iterations -- quadrotor / double-integrator subproblems solve cold in < 20)
    const int snap_have = a.snap[blockIdx.x];
    int snap_level = -1;
    {
        const double pd = a.prev_dev[blockIdx.x];
        // after a solve that ended at reduced accuracy (ALMOST_OPTIMAL: the gap stalled) the very fine level is not used: the last
        // iterates of such a solve are not well centred, and a start from them took 45 iterations + the cold repeat on the same
        // handful of problems in every launch (they set the launch time: the mean was 1 ... 5 iterations)
        const int lmax = a.status[blockIdx.x] == IPM_ALMOST ? IpmArgs::NWL - 2 : IpmArgs::NWL - 1;
#pragma unroll
        for (int l = IpmArgs::NWL - 1; l >= 0; l--)
            if (l > lmax) continue; else
            if (snap_level < 0 && pd <= a.warm_dev[l] && ((snap_have >> l) & 1) != 0 && (l > 0 || a.cold_iters[blockIdx.x] >= a.warm_min_cold)) snap_level = l;
    }
    const bool try_warm = a.warm_allowed != 0 && a.status[blockIdx.x] <= IPM_ALMOST && snap_level >= 0;
    if (snap_level < 0) snap_level = 0;
    const int snap_prev = a.snap[blockIdx.x];
    int snap_taken = 0, snap_keep = 0;
    double reg_cur = a.reg;   // static regularisation of this solve: escalated when a factorisation breaks down (below)
    bool warm = false;
    // attempt 0: warm start; 1: cold; 2: cold with the iterative refinement switched on from the first iteration (the
    // default refines only once relgap < ref_gap: on ~0.3 % of the rocket Monte-Carlo subproblems the unrefined early
    // directions leave the run stuck at a gap of 1e-4 ... 1 until the iteration limit -- the literal program's solver and the
    // scalar CPU restatement solve those instances, tests/test_failures_gpu.py; with refinement throughout this solver does too)
    bool robust = false;
    bool fell_back = false;   // the warm attempt from the very fine level failed and is being repeated from the next level
    for (int attempt = try_warm ? 0 : 1; attempt < 3; attempt++) {
    warm = attempt == 0;
    robust = attempt == 2;
    status = IPM_ITERLIM; best_it = 0; best_merit = 1e300; info_best[6] = 1e300; relgap_it = 1e300;
    s = W + wo.s; lam = W + wo.lam; r2 = W + wo.r2; el = W + wo.el;
    if (lane == 0) L->fail = 0;
    gsync();
    snap_taken = 0;
    // a warm solve that ends before it refreshes a snapshot (0 iterations on a converged reference) keeps the old one: without it
    // the NEXT solve of that problem was a cold one (45-50 iterations in a launch whose mean is 10)
    snap_keep = warm ? snap_prev : 0;
    if (warm) {
        const double* in[3] = {W + wo.sn_xi[snap_level], W + wo.sn_s[snap_level], W + wo.sn_lam[snap_level]};
        {
            const double* i1[1] = {in[0]};
            flat<1, 8>(XI, i1, [&](long i, const double(&v)[1]) { xi[i] = v[0]; });
        }
        {
            const double* i2[2] = {in[1], in[2]};
            flat<2, 8>(ROWS, i2, [&](long i, const double(&v)[2]) { s[i] = v[0]; lam[i] = v[1]; });
        }
        gsync();
    }
    for (it = warm ? 0 : -1; it <= a.max_iter; it++) {
        if (it < 0) {
            for (long i = lane; i < ROWS; i += 64) { w[i] = 1.0; rtil[i] = hneg[i]; r2[i] = 0.0; }
            for (long i = lane; i < XI; i += 64) { rx[i] = cv[i]; xi[i] = 0.0; dxi[i] = 0.0; rxe[i] = 0.0; }
            nt_identity();
            gsync();
        } else {
            // ---- residuals (+ scalings w = lam/s and the affine right-hand side r~z = rz - s in the same sweep) ----
            PH_T(1, (ipm2_ph_GT<M, WPE>(Pg, W, N, lam, rx)));
            PH_T(0, (ipm2_ph_G<M, WPE>(Pg, W, N, xi, gd)));
            double lrz = 0.0, nrz = 0.0, nrx = 0.0, pc = 0.0;
            gap = 0.0;
            OT_BEGIN();
            {
                const double* in[4] = {rx, xi, qd, cv};
                flat<4, 8>(XI, in, [&](long i, const double(&v)[4]) {
                    const double r_ = v[0] + v[2] * v[1] + v[3];
                    rx[i] = r_;
                    nrx += r_ * r_;
                    pc += 0.5 * v[2] * v[1] * v[1] + v[3] * v[1];
                });
            }
            {
                const double* in[4] = {gd, s, hneg, lam};
                flat<4, 8>(ROWS, in, [&](long i, const double(&v)[4]) {
                    const bool lv = !is_dead((int)i);
                    const double val = lv ? v[0] + v[1] + v[2] : 0.0;
                    rz[i] = val;
                    rtil[i] = val - v[1];
                    w[i] = (lv && !is_soc((int)i)) ? v[3] / v[1] : 1.0;
                    if (lv) { gap += v[1] * v[3]; lrz += v[3] * val; nrz += val * val; }
                });
            }
            gsync();
            gap = wave_sum(gap); lrz = wave_sum(lrz); nrz = wave_sum(nrz); nrx = wave_sum(nrx);
            if (!(warm && it == 0)) {   // warm-start snapshots of this solve: the first iterates with mu below the fine / the coarse
                // level (a warm solve refreshes them only after a step on the NEW problem: its start point is the old snapshot)
                const double mu_now = gap / deg;
#pragma unroll 1
                for (int q = IpmArgs::NWL - 1; q >= 0; q--) {
                    if ((snap_taken >> q) & 1) continue;
                    if (warm && q < snap_level) continue;      // a warm solve refreshes its own level and the finer ones only (it never comes
                                                               // from above a coarser one; the fall-back below relies on that snapshot)
                    const double lvl_mu = q == 3 ? a.warm_mu[3] : (q == 2 ? a.warm_mu[2] : (q == 1 ? a.warm_mu[1] : a.warm_mu[0]));   // (no dynamic index into the argument block)
                    static_assert(IpmArgs::NWL == 4, "level thresholds are selected by hand");
                    // a level is refreshed by an iterate within two decades below it only: a warm solve that starts far below a level
                    // leaves that snapshot alone (kept, snap_keep) -- otherwise the levels drift finer with every warm solve (mu = 1e-13
                    // under the 1e-10 label after eight launches) until a start is too close to the boundary even for a 1e-9 move
                    if (!(mu_now <= lvl_mu && mu_now >= 1e-2 * lvl_mu)) continue;
                    double* o_xi = W + wo.sn_xi[q]; double* o_s = W + wo.sn_s[q]; double* o_l = W + wo.sn_lam[q];
                    {
                        const double* i1[1] = {xi};
                        flat<1, 8>(XI, i1, [&](long i, const double(&v)[1]) { o_xi[i] = v[0]; });
                    }
                    {
                        const double* i2[2] = {s, lam};
                        flat<2, 8>(ROWS, i2, [&](long i, const double(&v)[2]) { o_s[i] = v[0]; o_l[i] = v[1]; });
                    }
                    snap_taken |= 1 << q;
                }
            }
            const double pcost = wave_sum(pc);
            const double dcost = pcost + lrz - gap;
            const double pres = sqrt(nrz) / nrm_h, dres = sqrt(nrx) / nrm_c;
            const double relgap = pcost < 0.0 ? gap / -pcost : (dcost > 0.0 ? gap / dcost : 1e300);
            relgap_it = relgap;
            const double merit = fmax(fmax(pres / a.feastol, dres / a.feastol), fmin(gap / a.abstol, relgap / a.reltol));
            const bool finite_ok = isfinite(merit) && (L->fail == 0);
            if (finite_ok && merit < best_merit) {
                best_merit = merit; best_it = it;
                const double* in[1] = {xi};
                flat<1, 8>(XI, in, [&](long i, const double(&v)[1]) { best[i] = v[0]; });
                info_best[0] = pcost + cost_const; info_best[1] = dcost + cost_const; info_best[2] = gap; info_best[3] = pres;
                info_best[4] = dres; info_best[5] = relgap; info_best[6] = merit;
                gsync();
            }
            if (!finite_ok) { status = IPM_NUMERR; break; }
            if (merit <= 1.0) { status = IPM_OPTIMAL; break; }
            if (it == a.max_iter) break;
            // a warm start that has not converged by now is abandoned.  From the very fine level a healthy solve needs <= 12 iterations
            // (p99 of the 4096 batch): 16 there, 45 elsewhere
            if (warm && it >= (snap_level == IpmArgs::NWL - 1 ? 16 : 45)) break;
            if (best_merit <= 1e3 && it - best_it >= a.stall) break;
            OT_END(0);
            PH_T(4, (ipm2_ph_nt<M, WPE>(Pg, W, N, s, lam)));
            OT_END(1);
            if (L->fail) { status = IPM_NUMERR; break; }
            mu = gap / deg;
        }
        PH_T(2, (ipm2_ph_factor<M, WPE>(Pg, W, N, reg_cur, w)));
        // A factorisation that breaks down (non-positive pivot) is repeated with 10x the static regularisation, which then stays for
        // the rest of this solve.  The default (1e-12, round 4) is 50x below the value that never broke down (5e-11): the end-game
        // of a solve -- gap 1e-5 -> 1e-8 with multipliers of 1e3 next to slacks of 1e-14 -- is limited by the accuracy of the
        // directions, and the smaller perturbation saves a third of the iterations of a warm-started solve (32 -> 21).
#pragma unroll 1
        for (int tr = 0; tr < 4 && L->fail; tr++) {
            reg_cur *= 10.0;
            gsync();
            if (lane == 0) L->fail = 0;
            gsync();
            ipm2_ph_factor<M, WPE>(Pg, W, N, reg_cur, w);
        }
        if (L->fail) { status = IPM_NUMERR; break; }
        // it < 0 (initial point, ECOS-style): phase 0 = primal point  min |G xi - h|^2 (+ xi'P xi), s = h - G xi ;
        //                                     phase 1 = dual point    min |lam|^2 s.t. P xi + G'lam + c = 0, lam = G xi_d
        const int nphase = 2;
        for (int phase = 0; phase < nphase; phase++) {
            if (it >= 0 && phase == 1) {
                // combined direction: r~z = rz - s + (sigma mu - ds_a dl_a)/lam ; cones: rz + W (lam~ \ d_s)
                OT_BEGIN();
                const double* in[5] = {rz, s, ds, dl, lam};
                flat<5, 8>(ROWS, in, [&](long i, const double(&v)[5]) {
                    if (is_soc((int)i)) return;
                    double val = v[0] - v[1];
                    if (!is_dead((int)i)) val += (sigma * mu - v[2] * v[3]) / v[4];
                    rtil[i] = val;
                });
                for (int idx = lane; idx < ncone; idx += 64) {
                    const int b0 = cone_base(idx);
                    const double* Wm = socW + (long)idx * 36;
                    double dsv_[4], dlv_[4], rzv[4], Wv[36];
#pragma unroll
                    for (int q = 0; q < 4; q++) { dsv_[q] = ds[b0 + q]; dlv_[q] = dl[b0 + q]; rzv[q] = rz[b0 + q]; }
#pragma unroll
                    for (int q = 0; q < 36; q++) Wv[q] = Wm[q];
                    const double* Wi = Wv + 16;
                    const double* lt = Wv + 32;
                    double u1[4], u2[4], dsv[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        double a1 = 0.0, a2 = 0.0;
#pragma unroll
                        for (int q2 = 0; q2 < 4; q2++) { a1 += Wi[q * 4 + q2] * dsv_[q2]; a2 += Wv[q * 4 + q2] * dlv_[q2]; }
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
                        for (int q2 = 0; q2 < 4; q2++) acc += Wv[q * 4 + q2] * uu[q2];
                        rtil[b0 + q] = rzv[q] + acc;
                    }
                }
                gsync();
                OT_END(2);
            }
            // ---- Newton solve + iterative refinement in augmented form ----
            // refinement only once the gap is small (the Newton system is well conditioned early on)
            const int nref_eff = (it < 0 || !(robust || relgap_it < a.ref_gap)) ? 0 : (a.nref > 0 ? a.nref : (robust ? 1 : 0));
            for (int rf = 0; rf <= nref_eff; rf++) {
                double *rt_ = rtil, *rx_ = rx, *ox = dxi, *og = gd, *ol = dl;
                if (it < 0 && phase == 0) { rx_ = rxe; ox = xi; }             // rhs (0, h)
                if (it < 0 && phase == 1) { rt_ = r2; og = ge; ol = el; }      // rhs (-c, 0)
                if (rf > 0) {
                    // -r1 = rx + P dxi + G'dl   (rxe) ;  -r2 = r~z + gd - W^2 dl   (r2)
                    PH_T(1, (ipm2_ph_GT<M, WPE>(Pg, W, N, dl, rxe)));
                    OT_BEGIN();
                    double n1 = 0.0, n2 = 0.0;   // squared norms of the two residual blocks
                    {
                        const double* in[4] = {rxe, qd, dxi, rx};
                        flat<4, 8>(XI, in, [&](long i, const double(&v)[4]) { const double r_ = v[0] + v[1] * v[2] + v[3]; rxe[i] = r_; n1 += r_ * r_; });
                    }
                    {
                        const double* in[4] = {rtil, gd, dl, w};
                        flat<4, 8>(ROWS, in, [&](long i, const double(&v)[4]) {
                            if (is_soc((int)i)) return;
                            const double r_ = is_dead((int)i) ? 0.0 : v[0] + v[1] - v[2] / v[3];
                            r2[i] = r_; n2 += r_ * r_;
                        });
                    }
                    for (int idx = lane; idx < ncone; idx += 64) {
                        const int b0 = cone_base(idx);
                        const double* Wm = socW + (long)idx * 36;
                        double dlv_[4], t1[4], Wv[16];
#pragma unroll
                        for (int q = 0; q < 16; q++) Wv[q] = Wm[q];
#pragma unroll
                        for (int q = 0; q < 4; q++) dlv_[q] = dl[b0 + q];
#pragma unroll
                        for (int q = 0; q < 4; q++) { double acc = 0.0; for (int q2 = 0; q2 < 4; q2++) acc += Wv[q * 4 + q2] * dlv_[q2]; t1[q] = acc; }
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {
                            double acc = 0.0;
#pragma unroll
                            for (int q = 0; q < 4; q++) acc += Wv[rr * 4 + q] * t1[q];
                            const double r_ = rtil[b0 + rr] + gd[b0 + rr] - acc;
                            r2[b0 + rr] = r_; n2 += r_ * r_;
                        }
                    }
                    gsync();
                    // adaptive: the correction solve is skipped when the direction already satisfies the Newton
                    // system to well below the feasibility tolerance (always the case for well-conditioned problems)
                    n1 = wave_sum(n1); n2 = wave_sum(n2);
                    OT_END(3);
                    if (sqrt(n1) <= a.ref_tol * a.feastol * nrm_c && sqrt(n2) <= a.ref_tol * a.feastol * nrm_h) break;
                    rt_ = r2; rx_ = rxe; ox = exi; og = ge; ol = el;
                }
                PH_T(3, (ipm2_ph_newton<M, WPE>(Pg, W, N, w, rt_, rx_, ox)));
                PH_T(6, (ipm2_ph_finish<M, WPE>(Pg, W, N, w, rt_, rx_, ox, og, ol)));
                if (rf > 0) {
                    OT_BEGIN();
                    {
                        const double* in[2] = {dxi, exi};
                        flat<2, 8>(XI, in, [&](long i, const double(&v)[2]) { dxi[i] = v[0] + v[1]; });
                    }
                    {
                        const double* in[4] = {dl, el, gd, ge};
                        flat<4, 8>(ROWS, in, [&](long i, const double(&v)[4]) { dl[i] = v[0] + v[1]; gd[i] = v[2] + v[3]; });
                    }
                    gsync();
                    OT_END(4);
                }
            }
            if (it < 0 && phase == 0) {
                for (long i = lane; i < ROWS; i += 64) s[i] = -(gd[i] + hneg[i]);
                gsync();
            } else if (it < 0) {
                // lam = G xi_d ; shift both into the cone
                for (long i = lane; i < ROWS; i += 64) lam[i] = ge[i];
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
                // ---- ds = -rz - G dxi and the largest feasible step for (s, ds), (lam, dl) in one sweep ----
                double am_s = 1e300, am_l = 1e300;   // largest steps keeping s and lam in the cone, separately
                OT_BEGIN();
                {
                    const double* in[5] = {rz, gd, s, lam, dl};
                    flat<5, 8>(ROWS, in, [&](long i, const double(&v)[5]) {
                        const double d = -v[0] - v[1];
                        ds[i] = d;
                        if (is_dead((int)i) || is_soc((int)i)) return;
                        if (d < 0.0) am_s = fmin(am_s, -v[2] / d);
                        if (v[4] < 0.0) am_l = fmin(am_l, -v[3] / v[4]);
                    });
                }
                for (int idx = lane; idx < ncone; idx += 64) {
                    const int b0 = cone_base(idx);
                    double sv[4], dsv_[4], lv_[4], dlv_[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) { sv[q] = s[b0 + q]; dsv_[q] = -rz[b0 + q] - gd[b0 + q]; lv_[q] = lam[b0 + q]; dlv_[q] = dl[b0 + q]; }
                    am_s = fmin(am_s, soc_step(sv, dsv_)); am_l = fmin(am_l, soc_step(lv_, dlv_));
                }
                am_s = wave_min(am_s); am_l = wave_min(am_l);
                const double am = fmin(am_s, am_l);
                gsync();
                OT_END(5);
                if (phase == 0) {
                    const double a_aff = fmin(1.0, am);
                    sigma = (1.0 - a_aff) * (1.0 - a_aff) * (1.0 - a_aff);
                } else {
                    // step: the new (s, lam) go to scratch row-vectors and the buffers are swapped once the
                    // iterate is verified interior (first trial almost always)
                    // Without a quadratic cost term the dual residual is linear in lam alone, so the primal pair
                    // (xi, s) and the multipliers may take different step lengths (a.split_step); otherwise one
                    // common step as in ECOS.
                    const bool split = a.split_step != 0 && !has_quad;
                    double alpha = fmin(1.0, 0.99 * (split ? am_s : am)), alpha_d = fmin(1.0, 0.99 * (split ? am_l : am));
                    for (int bt = 0; bt < 60; bt++) {
                        double mm_s = 1e300, mm_l = 1e300;
                        const double* in[4] = {s, ds, lam, dl};
                        flat<4, 8>(ROWS, in, [&](long i, const double(&v)[4]) {
                            const double sn = v[0] + alpha * v[1], ln = v[2] + alpha_d * v[3];
                            r2[i] = sn; el[i] = ln;
                            if (!is_dead((int)i) && !is_soc((int)i)) { mm_s = fmin(mm_s, sn); mm_l = fmin(mm_l, ln); }
                        });
                        for (int idx = lane; idx < ncone; idx += 64) {
                            const int b0 = cone_base(idx);
                            double t[4], u[4];
#pragma unroll
                            for (int q = 0; q < 4; q++) { t[q] = s[b0 + q] + alpha * ds[b0 + q]; u[q] = lam[b0 + q] + alpha_d * dl[b0 + q]; }
                            mm_s = fmin(mm_s, t[0] - sqrt(t[1] * t[1] + t[2] * t[2] + t[3] * t[3]));
                            mm_l = fmin(mm_l, u[0] - sqrt(u[1] * u[1] + u[2] * u[2] + u[3] * u[3]));
                        }
                        mm_s = wave_min(mm_s); mm_l = wave_min(mm_l);
                        if (mm_s > 0.0 && mm_l > 0.0) break;
                        if (split) { if (!(mm_s > 0.0)) alpha *= 0.8; if (!(mm_l > 0.0)) alpha_d *= 0.8; }
                        else { alpha *= 0.8; alpha_d = alpha; }
                        gsync();
                    }
                    { double* t_ = s; s = r2; r2 = t_; t_ = lam; lam = el; el = t_; }
                    const double* in[2] = {xi, dxi};
                    flat<2, 8>(XI, in, [&](long i, const double(&v)[2]) { xi[i] = v[0] + alpha * v[1]; });
                    gsync();
                    OT_END(6);
                }
            }
        }
    }
    if (it < 0) it = 0;
    iters_total += it;
    if (status != IPM_OPTIMAL) {
        // ECOS "reduced tolerances" -> ALMOST_OPTIMAL
        if (info_best[3] <= 1e-4 && info_best[4] <= 1e-4 && (info_best[2] <= 5e-5 || info_best[5] <= 5e-5)) status = IPM_ALMOST;
    }
    // a warm start that failed, or that ended at reduced accuracy with a primal / dual residual above the tolerance, is
    // repeated cold (a cold ALMOST_OPTIMAL exit has residuals at round-off: only the gap stalls)
    if (warm) {
        if (status == IPM_OPTIMAL || (status == IPM_ALMOST && info_best[3] <= a.feastol && info_best[4] <= a.feastol)) break;
        // a failed start from the very fine level is repeated from the next level before the cold repeat: that snapshot is untouched
        // (rule above) and 45 + 35 ... 60 iterations become 16 + ~10 (round 6; the rare failures set the launch time of the late launches)
        if (!fell_back && snap_level == IpmArgs::NWL - 1 && ((snap_prev >> (IpmArgs::NWL - 2)) & 1) != 0) {
            fell_back = true; snap_level = IpmArgs::NWL - 2; attempt = -1;
        }
    } else if (status <= IPM_ALMOST || robust) break;
    }   // attempt
    if (!warm && lane == 0) a.cold_iters[blockIdx.x] = it;      // iterations of the last attempt from a cold point (cold or robust)
    if (lane == 0) a.snap[blockIdx.x] = snap_taken | snap_keep;
    it = iters_total;
    PROF_ADD2(7, tick() - t_start_);
    // ---------------- result: best iterate ----------------
    for (long i = lane; i < (long)N * nz; i += 64) a.z_out[(long)blockIdx.x * N * nz + i] = best[i];
    if (lane < npa) a.p_out[(long)blockIdx.x * npa + lane] = PV(best, lane);
    if (lane == 0) {
        a.status[blockIdx.x] = status;
        a.iters[blockIdx.x] = it;
        for (int i = 0; i < 7; i++) a.info[(long)blockIdx.x * 8 + i] = info_best[i];
        a.info[(long)blockIdx.x * 8 + 7] = (double)best_it;
        if (a.prof) for (int i = 0; i < 8; i++) a.prof[(long)blockIdx.x * 8 + i] = L->prof[i];
    }
}

// WPE = waves per SIMD the kernel (and, through the attribute propagation, its phase functions) is compiled for:
//   WPE 1: up to 512 registers per lane, fastest single wave -- batches that cannot fill the chip twice;
//   WPE 2: 256 registers, two problems share a SIMD and hide each other's memory / LDS latency -- large batches.
template <class M, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void ipm2_solve_kernel(IpmArgs a, ExtractArgs ext)
{
    if (a.active != nullptr && a.active[blockIdx.x] == 0) return;
    Ipm2<M> S_;
    S_.a = a;
    S_.N = a.N;
    S_.lane = threadIdx.x;
    S_.Pg = a.slab + (long)blockIdx.x * a.slab_stride;
    S_.o = SP<M>::offsets(a.N);
    S_.wo = Ipm2Work<M>::offsets(a.N);
    S_.W = a.work + (long)blockIdx.x * a.work_stride;
    S_.L = ipm2_lds<M, WPE>();
    S_.ttrp = S_.Pg[S_.o.scal + 0];
    S_.cost_const = S_.Pg[S_.o.scal + 1];
    S_.template run<WPE>();
    // K4a in the tail of the wave that solved the problem (ext.xd != nullptr; ptr_kernels.hpp): the solution just written by
    // other lanes of this wave is read back from global memory, hence the fence
    if (ext.xd != nullptr) {
        __threadfence();
        __syncthreads();
        ptr_extract_body<M, WPE>(ext, blockIdx.x, threadIdx.x);
    }
}

}  // namespace scp
