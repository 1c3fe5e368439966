// TEST INFRASTRUCTURE (not product code): host build of the PRODUCT's generic conic solver sources
// (scptoolbox.jl_amd/csrc/conic_symbolic.hpp + conic_ipm.hpp), so that the numerics of the solver body -- which is
// plain `__host__ __device__` C++ -- can be checked on a machine without a GPU against the independent restatement
// oracle/ipm.py (tests/test_conic_cpu.py), and timed as the CPU leg of the generic path.  The product never loads this
// library: scptoolbox.jl_amd/conic.py binds libscp_mi355x.so only and fails without it.
//
// Same memory layout as on the device: every per-problem array interleaved across the batch.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/scp_conic.h"
#include "../scptoolbox.jl_amd/csrc/conic_ipm.hpp"
#include "../scptoolbox.jl_amd/csrc/conic_symbolic.hpp"

using namespace scp::conic;

// Multi-worker emulation of the device context (CONIC_HOST_WORKERS=n): n host threads play the worker waves of one
// problem, with a real barrier and shared reduction slots -- a missing barrier in the solver body shows up as a data
// race (wrong / irreproducible results) on the CPU, without a GPU.
struct ThreadShared {
    int nw;
    std::atomic<int> count{0};
    std::atomic<int> sense{0};
    std::vector<double> red;
    explicit ThreadShared(int n) : nw(n), red(n) {}
};
// Single worker that sums every item in the ORDER the device does when a workgroup owns one problem (CONIC_HOST_COOP=<workers of the
// device group, 1024>): groups of G lanes with strided terms and a butterfly of partial sums (conic_ipm.hpp, Solver::pfor_coop).
struct SerialCoopCtx {
    static constexpr bool COOP = true;
    static constexpr bool COOP_EMU = true;
    int vw = 1024;
    int coop_workers() const { return vw; }
    double gsum(double v, int) const { return v; }
    int wid() const { return 0; }
    int nw() const { return 1; }
    void barrier() const {}
    double sum(double v) const { return v; }
    double min(double v) const { return v; }
    bool any(bool v) const { return v; }
};

struct ThreadCtx {
    static constexpr bool COOP = false;
    static constexpr bool COOP_EMU = false;
    int coop_workers() const { return 1; }
    double gsum(double v, int) const { return v; }
    ThreadShared* sh;
    int w;
    mutable int local_sense = 0;
    int wid() const { return w; }
    int nw() const { return sh->nw; }
    void barrier() const
    {
        local_sense ^= 1;
        if (sh->count.fetch_add(1) == sh->nw - 1) { sh->count.store(0); sh->sense.store(local_sense); }
        else while (sh->sense.load() != local_sense) std::this_thread::yield();
    }
    double sum(double v) const
    {
        sh->red[w] = v; barrier();
        double acc = 0.0;
        for (int i = 0; i < sh->nw; i++) acc += sh->red[i];
        barrier();
        return acc;
    }
    double min(double v) const
    {
        sh->red[w] = v; barrier();
        double acc = sh->red[0];
        for (int i = 1; i < sh->nw; i++) acc = std::fmin(acc, sh->red[i]);
        barrier();
        return acc;
    }
    bool any(bool v) const { const double r = sum(v ? 1.0 : 0.0); return r > 0.0; }
};

static Csc make_csc(int nrow, int ncol, const int* p, const int* i)
{
    Csc M;
    M.nrow = nrow; M.ncol = ncol;
    if (p == nullptr) { M.p.assign(ncol + 1, 0); return M; }
    M.p.assign(p, p + ncol + 1);
    if (M.p[ncol] > 0) M.i.assign(i, i + M.p[ncol]);
    return M;
}

// Work profile of a schedule, per elimination level: [columns, row entries (pivot sums = forward substitution terms), longest
// row, L entries, operand pairs, longest pair list] -- what tools/order_survey.py prices the level-scheduled kernel's critical
// path with (a worker owns a whole row / a whole pair list), then the same with the long items cut into chunks:
// [6] critical path of the level's pivot / forward phase (longest short row, or longest chunk + chunks of a row), [7] the same
// for its entry phase.  prof[nlev][8]; returns nlev (prof may be NULL).
extern "C" int conic_host_schedule_profile(int n, int p, int m, int l, int ncones, const int* q, const int* Pp, const int* Pi,
                                           const int* Ap, const int* Ai, const int* Gp, const int* Gi, long long* prof, int cap)
{
    Symbolic S;
    try {
        const char* om = std::getenv("CONIC_HOST_ORDER");
        const std::string order = om ? om : "seq";
        const std::vector<int> qv(q, q + ncones);
        if (order == "nd" || order == "best")
            S = analyse_auto(n, p, m, l, qv, make_csc(n, n, Pp, Pi), make_csc(p, n, Ap, Ai), make_csc(m, n, Gp, Gi),
                             std::getenv("CONIC_HOST_WORKERS") ? std::atoi(std::getenv("CONIC_HOST_WORKERS")) : 256, order == "best",
                             nullptr, nullptr);
        else
            S = analyse(n, p, m, l, qv, make_csc(n, n, Pp, Pi), make_csc(p, n, Ap, Ai), make_csc(m, n, Gp, Gi), nullptr, false, ORDER_SEQUENTIAL);
    } catch (const std::exception&) {
        return -1;
    }
    const int nlev = (int)S.lev_p.size() - 1;
    if (!prof) return nlev;
    for (int lv = 0; lv < nlev && lv < cap; lv++) {
        long long* o = prof + 8LL * lv;
        o[0] = S.lev_p[lv + 1] - S.lev_p[lv]; o[1] = o[2] = 0;
        for (int t = S.lev_p[lv]; t < S.lev_p[lv + 1]; t++) {
            const int j = S.lev_cols[t];
            const long long len = S.row_p[j + 1] - S.row_p[j];
            o[1] += len; o[2] = std::max(o[2], len);
        }
        o[3] = S.lev_ent_p[lv + 1] - S.lev_ent_p[lv]; o[4] = o[5] = 0;
        for (int t = S.lev_ent_p[lv]; t < S.lev_ent_p[lv + 1]; t++) {
            const int e = S.lev_ent[t];
            const long long len = S.pair_p[e + 1] - S.pair_p[e];
            o[4] += len; o[5] = std::max(o[5], len);
        }
        o[6] = o[7] = 0;
        for (int t = S.lev_p[lv]; t < S.lev_p[lv + 1]; t++) {
            const int j = S.lev_cols[t];
            if (t < S.lev_p[lv] + S.lev_nshort[lv]) { o[6] = std::max<long long>(o[6], S.row_p[j + 1] - S.row_p[j]); continue; }
            long long mc = 0;
            for (int c = S.col_c0[t]; c < S.col_c1[t]; c++) mc = std::max<long long>(mc, S.rchunk_r1[c] - S.rchunk_r0[c]);
            o[6] = std::max<long long>(o[6], mc + (S.col_c1[t] - S.col_c0[t]));
        }
        for (int t = S.lev_ent_p[lv]; t < S.lev_ent_p[lv + 1]; t++) {
            const int e = S.lev_ent[t];
            if (t < S.lev_ent_p[lv] + S.lev_ent_nshort[lv]) { o[7] = std::max<long long>(o[7], S.pair_p[e + 1] - S.pair_p[e]); continue; }
            long long mc = 0;
            for (int c = S.ent_c0[t]; c < S.ent_c1[t]; c++) mc = std::max<long long>(mc, S.echunk_q1[c] - S.echunk_q0[c]);
            o[7] = std::max<long long>(o[7], mc + (S.ent_c1[t] - S.ent_c0[t]));
        }
    }
    return nlev;
}

extern "C" int conic_host_solve(int n, int p, int m, int l, int ncones, const int* q, const int* Pp, const int* Pi,
                                const int* Ap, const int* Ai, const int* Gp, const int* Gi, const int* perm, int B,
                                const double* c, const double* b, const double* hvec, const double* Gx, const double* Ax,
                                const double* Px, unsigned shared_mask, const scp_conic_opts* opts, double* x, double* y,
                                double* z, double* s, int32_t* status, int32_t* iters, double* info, long long* stats)
{
    Symbolic S;
    std::vector<int> ctype(ncones > 0 ? ncones : 1, 0), cexp(ncones > 0 ? ncones : 1, -1);
    int nexp = 0;
    try {
        const char* om = std::getenv("CONIC_HOST_ORDER");
        const std::string order = om ? om : "seq";
        std::vector<int> qv(q, q + ncones);
        for (int c = 0; c < ncones; c++) if (qv[c] == -3) { qv[c] = 3; ctype[c] = 1; cexp[c] = nexp++; }      // exponential cones
        // "nd" / "best": what Engine::create does (analyse_auto: the cheapest dissection; "best" may also keep the sequential order)
        if (perm == nullptr && (order == "nd" || order == "best"))
            S = analyse_auto(n, p, m, l, qv, make_csc(n, n, Pp, Pi), make_csc(p, n, Ap, Ai), make_csc(m, n, Gp, Gi),
                             std::getenv("CONIC_HOST_WORKERS") ? std::atoi(std::getenv("CONIC_HOST_WORKERS")) : 256, order == "best",
                             nullptr, nullptr);
        else
            S = analyse(n, p, m, l, qv, make_csc(n, n, Pp, Pi), make_csc(p, n, Ap, Ai), make_csc(m, n, Gp, Gi), perm,
                        std::getenv("CONIC_FREE_ORDER") != nullptr, ORDER_SEQUENTIAL);
    } catch (const std::exception&) {
        return 1;
    }
    const CsrView Gr = csr_view(S.G);
    std::vector<int2_> pairs(S.pair_a.size());
    for (size_t i = 0; i < pairs.size(); i++) { pairs[i].a = S.pair_a[i]; pairs[i].b = S.pair_b[i]; }
    std::vector<long long> pair_p(S.pair_p.begin(), S.pair_p.end());
    Sched D;
    D.n = n; D.p = p; D.m = m; D.l = l; D.nk = S.nk; D.ncone = ncones;
    D.nnzG = S.G.nnz(); D.nnzGt = S.Gt.nnz(); D.nnzA = S.A.nnz(); D.nnzP = S.P.nnz(); D.nnzL = S.Lp[S.nk];
    D.njob = (int)S.job_gt0.size(); D.nlp = (int)S.lp_gt.size();
    D.q = S.q.data(); D.cone_off = S.cone_off.data(); D.ctype = ctype.data(); D.cexp = cexp.data(); D.nexp = nexp;
    D.Gp = S.G.p.data(); D.Gi = S.G.i.data(); D.Gr_p = Gr.p.data(); D.Gr_j = Gr.j.data(); D.Gr_pos = Gr.pos.data();
    D.Gtp = S.Gt.p.data(); D.Gti = S.Gt.i.data(); D.Gtr_p = S.Gtr.p.data(); D.Gtr_j = S.Gtr.j.data(); D.Gtr_pos = S.Gtr.pos.data();
    D.Ap = S.A.p.data(); D.Ai = S.A.i.data(); D.Ar_p = S.Ar.p.data(); D.Ar_j = S.Ar.j.data(); D.Ar_pos = S.Ar.pos.data();
    D.Pf_p = S.Pfull.p.data(); D.Pf_j = S.Pfull.j.data(); D.Pf_pos = S.Pfull.pos.data();
    D.kk_p = S.kk_p.data(); D.kk_src = S.kk_src.data(); D.kk_idx = S.kk_idx.data(); D.kk_col = S.kk_col.data();
    D.kk_long = S.kk_long.data(); D.nkk_long = (int)S.kk_long.size(); D.kk_long_thr = Symbolic::KK_LONG;
    D.job_gt0 = S.job_gt0.data(); D.job_cone = S.job_cone.data(); D.job_src_p = S.job_src_p.data();
    D.job_src_row = S.job_src_row.data(); D.job_src_g = S.job_src_g.data(); D.lp_gt = S.lp_gt.data(); D.lp_g = S.lp_g.data();
    D.perm = S.perm.data();
    D.Lp = S.Lp.data(); D.Li = S.Li.data(); D.l_src = S.l_src.data(); D.l_src_idx = S.l_src_idx.data();
    D.d_src = S.d_src.data(); D.d_src_idx = S.d_src_idx.data(); D.d_kind = S.d_kind.data();
    D.pair_p = pair_p.data(); D.pairs = pairs.data();
    D.row_p = S.row_p.data(); D.row_k = S.row_k.data(); D.row_pos = S.row_pos.data();
    D.nlev = (int)S.lev_p.size() - 1; D.nrlev = (int)S.rlev_p.size() - 1;
    D.lev_p = S.lev_p.data(); D.lev_cols = S.lev_cols.data(); D.lev_ent_p = S.lev_ent_p.data(); D.lev_ent = S.lev_ent.data();
    D.ent_col = S.ent_col.data(); D.rlev_p = S.rlev_p.data(); D.rlev_cols = S.rlev_cols.data();
    std::vector<long long> eq0(S.echunk_q0.begin(), S.echunk_q0.end()), eq1(S.echunk_q1.begin(), S.echunk_q1.end());
    D.max_chunks = S.max_chunks;
    D.lev_nshort = S.lev_nshort.data(); D.rchunk_p = S.rchunk_p.data(); D.rchunk_r0 = S.rchunk_r0.data(); D.rchunk_r1 = S.rchunk_r1.data();
    D.col_c0 = S.col_c0.data(); D.col_c1 = S.col_c1.data();
    D.lev_ent_nshort = S.lev_ent_nshort.data(); D.echunk_p = S.echunk_p.data(); D.ent_c0 = S.ent_c0.data(); D.ent_c1 = S.ent_c1.data();
    D.echunk_q0 = eq0.data(); D.echunk_q1 = eq1.data();
    if (stats) { stats[0] = D.nnzL; stats[1] = S.flops; stats[2] = D.nk; stats[3] = D.nnzGt; stats[4] = S.nd_depth; stats[5] = D.nlev; stats[6] = D.nrlev;
        long mx = 0; for (int j = 0; j < S.nk; j++) mx = std::max<long>(mx, S.row_p[j + 1] - S.row_p[j]);
        stats[7] = mx; }
    if (std::getenv("CONIC_HOST_PERM_HASH")) {      // (test aid: the elimination order as one number, to compare two builds of the ordering code)
        unsigned long long hsh = 1469598103934665603ULL;
        for (int v : S.perm) { hsh ^= (unsigned long long)(unsigned)v; hsh *= 1099511628211ULL; }
        std::fprintf(stderr, "CONIC_HOST_PERM_HASH %016llx nk %d nnzL %d\n", hsh, S.nk, (int)S.Li.size());
    }
    if (B <= 0) return 0;

    Opts o = default_opts();
    if (opts) {
        o.max_iter = opts->max_iter; o.feastol = opts->feastol; o.abstol = opts->abstol; o.reltol = opts->reltol;
        o.reg = opts->reg; o.dyn_eps = opts->dyn_eps; o.dyn_delta = opts->dyn_delta; o.nref = opts->nref;
        o.ref_tol = opts->ref_tol; o.step = opts->step;
    }
    o.fine = 0;
    if (!(o.reg >= 0.0)) { o.reg = auto_reg(S.n_free, S.n, S.m, S.q.empty() && S.P.i.empty(), nexp > 0); o.fine = o.reg < 1e-9; }
    const long BS = B;
    // inputs: [len, B] column-major -> interleaved [len][BS]
    auto interleave = [&](const double* src, long len, bool shared) {
        std::vector<double> v((size_t)std::max<long>(len, 1) * (shared ? 1 : BS), 0.0);
        if (len == 0) return v;
        if (shared) { std::memcpy(v.data(), src, sizeof(double) * len); return v; }
        for (long t = 0; t < B; t++) for (long e = 0; e < len; e++) v[e * BS + t] = src[t * len + e];
        return v;
    };
    std::vector<double> ci = interleave(c, n, shared_mask & SCP_CONIC_SHARED_C), bi = interleave(b, p, shared_mask & SCP_CONIC_SHARED_B),
                        hi = interleave(hvec, m, shared_mask & SCP_CONIC_SHARED_H), Gi_ = interleave(Gx, D.nnzG, shared_mask & SCP_CONIC_SHARED_G),
                        Ai_ = interleave(Ax, D.nnzA, shared_mask & SCP_CONIC_SHARED_A), Pi_ = interleave(Px, D.nnzP, shared_mask & SCP_CONIC_SHARED_P);
    const long nk = D.nk;
    const long work_len = D.nnzGt + 2L * D.nnzL + nk + 5 * nk + D.max_chunks + 6L * m + ncones + 9L * nexp + n + p;
    std::vector<double> work((size_t)work_len * BS, 0.0), xs((size_t)std::max(n, 1) * BS), ys((size_t)std::max(p, 1) * BS),
        zs((size_t)std::max(m, 1) * BS), ss((size_t)std::max(m, 1) * BS);
    const int workers = std::getenv("CONIC_HOST_WORKERS") ? std::atoi(std::getenv("CONIC_HOST_WORKERS")) : 1;
#pragma omp parallel for schedule(dynamic) if (workers <= 1)
    for (int t = 0; t < B; t++) {
        Prob Q;
        auto cb = [&](std::vector<double>& v, bool shared) { return shared ? CBV{v.data(), 1} : CBV{v.data() + t, BS}; };
        Q.c = cb(ci, shared_mask & SCP_CONIC_SHARED_C); Q.b = cb(bi, shared_mask & SCP_CONIC_SHARED_B);
        Q.h = cb(hi, shared_mask & SCP_CONIC_SHARED_H); Q.Gx = cb(Gi_, shared_mask & SCP_CONIC_SHARED_G);
        Q.Ax = cb(Ai_, shared_mask & SCP_CONIC_SHARED_A); Q.Px = cb(Pi_, shared_mask & SCP_CONIC_SHARED_P);
        Q.x = BV{xs.data() + t, BS}; Q.y = BV{ys.data() + t, BS}; Q.z = BV{zs.data() + t, BS}; Q.s = BV{ss.data() + t, BS};
        double* w = work.data();
        auto take = [&](long len) { BV v{w + t, BS}; w += len * BS; return v; };
        Q.Gt = take(D.nnzGt); Q.Lx = take(D.nnzL); Q.Ux = take(D.nnzL); Q.Dinv = take(nk);
        Q.rhs = take(nk); Q.sol = take(nk); Q.res = take(nk); Q.cor = take(nk); Q.tmp = take(nk);
        Q.part = take(D.max_chunks);
        Q.lam = take(m); Q.wsc = take(m); Q.ds = take(m); Q.dz = take(m); Q.corr = take(m); Q.rz = take(m);
        Q.eta = take(ncones + 9L * nexp); Q.rx = take(n); Q.ry = take(p);
        Result R;
        const int coop = std::getenv("CONIC_HOST_COOP") ? std::atoi(std::getenv("CONIC_HOST_COOP")) : 0;
        if (workers <= 1 && coop > 1) {
            SerialCoopCtx cx; cx.vw = coop;
            Solver<SerialCoopCtx> sv(D, Q, o, cx);
            R = sv.run();
        } else if (workers <= 1) {
            SerialCtx cx;
            Solver<SerialCtx> sv(D, Q, o, cx);
            R = sv.run();
        } else {
            ThreadShared sh(workers);
            std::vector<std::thread> th;
            std::vector<Result> rs(workers);
            for (int w = 0; w < workers; w++)
                th.emplace_back([&, w]() {
                    ThreadCtx cx; cx.sh = &sh; cx.w = w;
                    Solver<ThreadCtx> sv(D, Q, o, cx);
                    rs[w] = sv.run();
                });
            for (auto& t_ : th) t_.join();
            R = rs[0];
            for (int w = 1; w < workers; w++)
                if (rs[w].status != R.status || rs[w].iters != R.iters || rs[w].pcost != R.pcost) R.status = 99;   // workers disagree
        }
        if (status) status[t] = R.status;
        if (iters) iters[t] = R.iters;
        if (info) {
            double* io = info + 8L * t;
            io[0] = R.pcost; io[1] = R.dcost; io[2] = R.gap; io[3] = R.pres; io[4] = R.dres; io[5] = R.relgap;
            io[6] = R.nreg; io[7] = R.nrefine;
        }
        if (x) for (long e = 0; e < n; e++) x[(long)t * n + e] = xs[e * BS + t];
        if (y) for (long e = 0; e < p; e++) y[(long)t * p + e] = ys[e * BS + t];
        if (z) for (long e = 0; e < m; e++) z[(long)t * m + e] = zs[e * BS + t];
        if (s) for (long e = 0; e < m; e++) s[(long)t * m + e] = ss[e * BS + t];
    }
    return 0;
}
