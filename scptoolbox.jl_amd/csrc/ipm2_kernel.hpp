// K3: batched stage-structured primal-dual interior-point solver for the reduced PTR subproblem
// (replaces JuMP `optimize!` -> ECOS, src/parser/program.jl:419-424 / src/solvers/scp.jl:942-950).
//
// Same algorithm as oracle/ipm_struct.py (see that file and DESIGN.md sections 2 and 4); this is the device
// implementation:
//   * one wavefront per problem, the whole IPM inside one launch; two kernel variants (WPE = 1 / 2 waves per SIMD,
//     512 / 256 registers) selected by batch size;
//   * every sweep over the horizon stages the part of the node's contiguous STAGE RECORD (csrc/stage_problem.hpp) it
//     reads, the row / primal records and the packed FACTOR RECORD through LDS with coalesced, unconditional
//     (clamped-index) loads, software-prefetched one node ahead into registers;
//   * the horizon sweeps (G, G', factor, solve, direction) are separate non-inlined device functions, and inside each
//     the first and last node are peeled off so that the hot loop holds one instantiation and no node-type predicates
//     (both for the sake of the register allocator);
//   * all inner loops have compile-time bounds (model dimensions are template constants);
//   * the block factorisation keeps EXPLICIT inverses of the small Cholesky factors, so the four triangular solves per
//     node and per right-hand side become dense mat-vecs (a triangular solve is a longer dependency chain for one wave);
//   * row-vector passes are flat, batched sweeps (8 loads in flight per lane);
//   * the direction pass fuses G*dxi, the epigraph-variable recovery, the multiplier recovery and ds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "ipm_kernel.hpp"   // IpmArgs, wave_* helpers, status codes
#include "stage_problem.hpp"
#include "ptr_kernels.hpp"   // ExtractArgs, ptr_extract_body (the fused tail of ipm2_solve_kernel)

namespace scp {

// Phase counters (scp_debug_get_ipm_profile) cost registers and s_memrealtime reads, so they are compiled only into
// the diagnostic builds: -DSCP_IPM_PROF (phases: G, G', factor, rhs+fwd, bwd, arrow, finish, total) or
// -DSCP_FACTOR_PROF (sub-phases of factor_stage: A, chol Sz, factor total, Y, Snu, chol Snu, X; currently
// miscompiled by hipcc 7.2 in the phase-function build -- not part of the Makefile).  The production library reports zeros.
#if defined(SCP_FACTOR_PROF)
#define SCP_TICK() ((long long)wall_clock64())
#define FPROF_BEGIN() long long fp_t_ = tick()
#define FPROF(i) do { const long long n_ = tick(); fprof_[i] += n_ - fp_t_; fp_t_ = n_; } while (0)
#define PROF_ADD(i, v) ((void)0)
#define PROF_ADD2(i, v) do { if (lane == 0) L->prof[i] += (v); } while (0)
#elif defined(SCP_IPM_PROF_CALLER)
#define SCP_TICK() ((long long)wall_clock64())
#define FPROF_BEGIN() ((void)0)
#define FPROF(i) ((void)0)
#define PROF_ADD(i, v) ((void)(v))
#define PROF_ADD2(i, v) do { if ((i) == 7 && lane == 0) L->prof[i] += (v); } while (0)
#elif defined(SCP_IPM_PROF_OTHER)
// the light passes of run() itself (everything the phase clocks of SCP_IPM_PROF report as "other"): slots 0 residual passes + snapshots,
// 1 nt_update, 2 combined right-hand side, 3 refinement residuals, 4 refinement updates, 5 step-length pass, 6 step trial + update
#define SCP_TICK() ((long long)wall_clock64())
#define FPROF_BEGIN() ((void)0)
#define FPROF(i) ((void)0)
#define PROF_ADD(i, v) ((void)(v))
#define PROF_ADD2(i, v) do { if ((i) == 7 && lane == 0) L->prof[i] += (v); } while (0)
#define OT_BEGIN() long long ot_t_ = (long long)wall_clock64()
#define OT_END(i) do { const long long n_ = (long long)wall_clock64(); if (lane == 0) L->prof[i] += n_ - ot_t_; ot_t_ = n_; } while (0)
#elif defined(SCP_IPM_PROF)
#define SCP_TICK() ((long long)wall_clock64())
#define FPROF_BEGIN() ((void)0)
#define FPROF(i) ((void)0)
#define PROF_ADD(i, v) do { if (lane == 0) L->prof[i] += (v); } while (0)
#define PROF_ADD2(i, v) do { if (lane == 0) L->prof[i] += (v); } while (0)
#else
#define SCP_TICK() (0LL)
#define FPROF_BEGIN() ((void)0)
#define FPROF(i) ((void)0)
#define PROF_ADD(i, v) ((void)(v))
#define PROF_ADD2(i, v) ((void)(v))
#endif
#ifndef OT_BEGIN
#define OT_BEGIN() ((void)0)
#define OT_END(i) ((void)0)
#endif

template <class M>
struct Ipm2Work {
    using S = SP<M>;
    __host__ __device__ static long XI(int N) { return (long)N * (S::nz + S::AS) + S::npa + S::AG; }
    __host__ __device__ static long ROWS(int N) { return (long)N * S::RS + S::RG; }
    // per-node factor record: [Li tri(nz) | Lni tri(MM) | X MM*nz | Y nz*MM | row coefficients MM*2]
    // Packed per node type: MM = number of nu-rows of the node (MNU at the two boundary nodes, MNU_MID inside).
    // Only the first f_used(MM) doubles of a record are ever written / read (327 of 592 for a rocket mid node).
    // the two inverse Cholesky factors are lower triangular and stored packed by rows: entry (i, j<=i) at i(i+1)/2 + j
    __host__ __device__ static constexpr int tri(int n) { return n * (n + 1) / 2; }
    __host__ __device__ static constexpr int f_lni(int) { return tri(S::nz); }
    __host__ __device__ static constexpr int f_x(int MM) { return tri(S::nz) + tri(MM); }
    __host__ __device__ static constexpr int f_y(int MM) { return f_x(MM) + MM * S::nz; }
    __host__ __device__ static constexpr int f_cf(int MM) { return f_y(MM) + S::nz * MM; }
    __host__ __device__ static constexpr int f_used(int MM) { return f_cf(MM) + 2 * MM; }
    static constexpr int FR = (f_used(S::MNU) + 7) & ~7;
    static constexpr int HS = (S::nz * S::nz + 1) & ~1;   // stride of an H0 block
    struct Off {
        long xi, dxi, rx, exi, best, rxe, cv, qd;                  // xi-vectors
        long s, lam, rz, w, rtil, ds, dl, gd, r2, el, hneg, ge;    // row-vectors
        long socW;                                                // [N][nsoc][36]: W(16) Wi(16) lamt(4)
        long F, C0, Ycz, Ycnu, fb, ft, nuv;                       // newton (F: per-node factor records)
        long tl;                                                  // [N][4 nsoc]: W^-1 (W^-1 rtil) of the cone rows (newton_rhs)
        long H0;                                                  // [N][HS]: chain-independent part of Sz_k (factor_pre)
        long sn_xi[IpmArgs::NWL], sn_s[IpmArgs::NWL], sn_lam[IpmArgs::NWL];   // warm-start snapshots (level 0 coarse ... NWL - 1 very fine): xi | s | lam
        long total;
    };
    __host__ __device__ static Off offsets(int N)
    {
        Off o;
        long c = 0;
        auto take = [&](long n) { long r = c; c += (n + 7) & ~7L; return r; };
        const long xi = XI(N), rows = ROWS(N);
        o.xi = take(xi); o.dxi = take(xi); o.rx = take(xi); o.exi = take(xi); o.best = take(xi); o.rxe = take(xi);
        o.cv = take(xi); o.qd = take(xi);
        o.s = take(rows); o.lam = take(rows); o.rz = take(rows); o.w = take(rows); o.rtil = take(rows);
        o.ds = take(rows); o.dl = take(rows); o.gd = take(rows); o.r2 = take(rows); o.el = take(rows);
        o.hneg = take(rows); o.ge = take(rows);
        o.socW = take((long)N * (S::nsoc > 0 ? S::nsoc : 1) * 36);
        o.F = take((long)N * FR);
        o.C0 = take((long)N * S::nz * S::npa);
        o.Ycz = take((long)N * S::nz * S::npa); o.Ycnu = take((long)N * S::MNU * S::npa);
        o.fb = take((long)N * S::nz); o.ft = take((long)N * S::MNU);
        o.nuv = take((long)N * S::MNU);
        o.tl = take((long)N * 4 * (S::nsoc > 0 ? S::nsoc : 1));
        o.H0 = take((long)N * HS);
        for (int q = 0; q < IpmArgs::NWL; q++) { o.sn_xi[q] = take(xi); o.sn_s[q] = take(rows); o.sn_lam[q] = take(rows); }
        o.total = c;
        return o;
    }
};

template <class M>
struct Ipm2 {
    using S = SP<M>;
    using WK = Ipm2Work<M>;
    static constexpr int nx = S::nx, nu = S::nu, np = S::np, npa = S::npa, nz = S::nz, ns = S::ns, nl = S::nl,
                         nsoc = S::nsoc, ml = S::ml, ng = S::ng, nic = S::nic, ntc = S::ntc, nbc = S::nbc, RS = S::RS,
                         RG = S::RG, AS = S::AS, AG = S::AG, MNU = S::MNU, MMID = S::MNU_MID, SR = S::SR, GR = S::GR;
    static_assert(nz <= 16 && MNU <= 16, "dense blocks must fit one DPP row (16 lanes)");
    static constexpr int NPRE = (SR + 63) / 64;
    static constexpr int FR = WK::FR, NPREF = (FR + 63) / 64, HS = WK::HS;
    static constexpr int NSOC1 = nsoc > 0 ? nsoc : 1;
    // Chain sweeps (forward / backward substitution over the horizon) read the factor record of a node in a PADDED PER-LANE layout:
    // slot s of lane l at Fpad[s * 16 + l], zeros where the triangles have none -- one unconditional LDS read per register, no index
    // clamps or selects (round 6; before: packed record + `q < lane ? v : 0` per element).  Slots: A (nz) | dA | B (nz) | C (MNU) | dC |
    // D (MNU); forward: A = row l of Lz, B = column l of Y, C = row l of Ln, D = column l of X; backward: A = column l of Lz,
    // B = row l of X, C = column l of Ln, D = row l of Y; dA, dC = reciprocal pivots.  The packed record (global memory, `Ipm2Work`)
    // is scattered into it with a per-lane offset table built once per sweep (pad_map).
    static constexpr int FS_A = 0, FS_DA = nz, FS_B = nz + 1, FS_C = 2 * nz + 1, FS_DC = 2 * nz + 1 + MNU, FS_D = 2 * nz + 2 + MNU,
                         FS_RB = 2 * nz + 2 + 2 * MNU, FS_RT = FS_RB + 1,          // the node's two right-hand-side entries
                         FSLOTS = 2 * nz + 4 + 2 * MNU, FPAD = FSLOTS * 16 + 64;   // + 64: dump slots for packed entries nobody reads
    struct LdsStage {
        double Pk[SR];           // stage record of the current node
        double Sz[nz * nz], Snu[MNU * MNU];
        double F[FR];            // factor record of the current node: Li | Lni | X | Y
    };
    struct Lds {
        union {
            LdsStage st;             // node-parallel sweeps and the factorisation
            double Fpad[FPAD];       // chain sweeps
        };
        double G[GR];            // global record
        double Ep[nx * nz];      // E of the previous node
        double zk[nz], zn[nz], ak[AS], pv[npa], ga[AG];
        double r0[RS], r1[RS];   // row staging
        double g0[RG], g1[RG];   // global-row staging
        double dcur[nx], dprev[nx];
        double soc[NSOC1 * 36];
        double Ysoc[4 * NSOC1 * nz];
        double Cz[nz * npa], cb[nz * npa], ct[MNU * npa];
        double thp[MNU], nuk[MNU];
        double arow[RS];         // main part of G*dxi per row
        double tmp[64];
        double spL[npa * npa];   // Cholesky factor of the arrow Schur complement (factor -> newton_solve)
        long long prof[8];       // phase counters (diagnostic builds), written by lane 0
        int fail;
    };

    int N, lane;
    const double* Pg;  // slab of this problem (global)
    typename S::Off o;
    typename WK::Off wo;
    double* W;
    Lds* L;
    IpmArgs a;
    double ttrp, cost_const;
#ifdef SCP_FACTOR_PROF
    long long fprof_[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // sub-phase ticks of this factor() call, flushed to LDS at its end
#endif
    double pre[NPRE];
    double preF[NPREF];
    static constexpr int NROWR = (RS + 63) / 64, NSOCR = (NSOC1 * 36 + 63) / 64;
    double pR0[NROWR], pR1[NROWR], pS[NSOCR], pZ, pA, pN, pB1, pB2;

    __device__ __forceinline__ long long tick() const { return SCP_TICK(); }
    // lsync: ordering point for LDS traffic inside the (single) wave.  LDS instructions of one wave execute
    // in issue order, so no s_barrier and -- crucially -- no `s_waitcnt vmcnt(0)` is needed: a __syncthreads()
    // here would drain the software prefetch of the next node and expose a full HBM/L2 round trip per node.
    __device__ __forceinline__ void lsync() const { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    // gsync: global-memory read-after-write across lanes (pass boundaries only)
    __device__ __forceinline__ void gsync() const { __syncthreads(); }
    __device__ __forceinline__ void sync() const { lsync(); }
    // Batched flat sweep over n contiguous doubles of NA arrays: every lane issues U*NA loads before the first
    // use, so a sweep costs ~n/(64 U) memory round trips instead of n/64 (one wave has nothing else to hide latency).
    template <int NA, int U, class F>
    __device__ __forceinline__ void flat(long n, const double* const (&p)[NA], F&& f) const
    {
        for (long base = lane; base < n; base += 64L * U) {
            double v[U][NA];
#pragma unroll
            for (int u = 0; u < U; u++) {
                long idx = base + 64L * u;
                idx = idx < n ? idx : n - 1;
#pragma unroll
                for (int q = 0; q < NA; q++) v[u][q] = p[q][idx];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const long idx = base + 64L * u;
                if (idx < n) f(idx, v[u]);
            }
        }
    }
    // ---------------- vector accessors ----------------
    __device__ __forceinline__ double& Z(double* v, int k, int j) const { return v[(long)k * nz + j]; }
    __device__ __forceinline__ double& AUX(double* v, int k, int i) const { return v[(long)N * nz + (long)k * AS + i]; }
    __device__ __forceinline__ double& PV(double* v, int j) const { return v[(long)N * (nz + AS) + j]; }
    __device__ __forceinline__ double& GAUX(double* v, int i) const { return v[(long)N * (nz + AS) + npa + i]; }
    __device__ __forceinline__ double& ROW(double* v, int k, int r) const { return v[(long)k * RS + r]; }
    __device__ __forceinline__ double& GROW(double* v, int r) const { return v[(long)N * RS + r]; }
    __device__ __forceinline__ bool live(int k, int r) const { return !(k == N - 1 && r < 2 * nx); }
    __device__ __forceinline__ int mnu(int k) const { return (k == 0 || k == N - 1) ? MNU : MMID; }

    // ---------------- staging helpers ----------------
    __device__ __forceinline__ void prefetch(int k)
    {
        const double* src = Pg + (long)k * SR;
#pragma unroll
        for (int i = 0; i < NPRE; i++) { const int idx = lane + 64 * i; pre[i] = src[idx < SR ? idx : SR - 1]; }   // unconditional (clamped) loads: no exec-mask branches
    }
    // partial refresh of the staged stage record: [LO, HI) only (sweeps that do not read the rest)
    template <int LO, int HI>
    __device__ __forceinline__ void prefetch_r(int k)
    {
        const double* src = Pg + (long)k * SR;
#pragma unroll
        for (int i = 0; i < (HI - LO + 63) / 64; i++) { const int idx = LO + lane + 64 * i; pre[i] = src[idx < HI ? idx : HI - 1]; }
    }
    template <int LO, int HI>
    __device__ __forceinline__ void commit_r()
    {
#pragma unroll
        for (int i = 0; i < (HI - LO + 63) / 64; i++) { const int idx = LO + lane + 64 * i; if (idx < HI) L->st.Pk[idx] = pre[i]; }
    }
    // the parameter columns only (Fp rows and Kp rows: everything Ft() reads) -- one load per lane
    static_assert(nx * npa + ml * npa <= 64, "parameter columns must fit one load per lane");
    __device__ __forceinline__ void prefetch_ft(int k)
    {
        const double* src = Pg + (long)k * SR;
        const int i0 = lane < nx * npa ? S::O_FP + lane : (lane < nx * npa + ml * npa ? S::O_KP + lane - nx * npa : S::O_KP);
        pre[0] = src[i0];
    }
    __device__ __forceinline__ void commit_ft()
    {
        if (lane < nx * npa) L->st.Pk[S::O_FP + lane] = pre[0];
        else if (lane < nx * npa + ml * npa) L->st.Pk[S::O_KP + lane - nx * npa] = pre[0];
    }
    __device__ __forceinline__ void commit()
    {
#pragma unroll
        for (int i = 0; i < NPRE; i++) { const int idx = lane + 64 * i; if (idx < SR) L->st.Pk[idx] = pre[i]; }
    }
    // factor-record staging: a mid node uses only the first FU_MID doubles of its record; the loads beyond that
    // are issued only for the two boundary nodes (wave-uniform branch)
    static constexpr int FU_MID = WK::f_used(MMID), FU_BND = WK::f_used(MNU);
    static constexpr int NPREF_MID = (FU_MID + 63) / 64;
    __device__ __forceinline__ bool bnd(int k) const { return k == 0 || k == N - 1; }
    __device__ __forceinline__ void prefetchF(int k)
    {
        const double* src = W + wo.F + (long)k * FR;
#pragma unroll
        for (int i = 0; i < NPREF_MID; i++) { const int idx = lane + 64 * i; preF[i] = src[idx < FR ? idx : FR - 1]; }
        if (bnd(k)) {
#pragma unroll
            for (int i = NPREF_MID; i < NPREF; i++) { const int idx = lane + 64 * i; preF[i] = src[idx < FR ? idx : FR - 1]; }
        }
    }
    __device__ __forceinline__ void commitF()
    {
#pragma unroll
        for (int i = 0; i < NPREF; i++) { const int idx = lane + 64 * i; if (idx < FR) L->st.F[idx] = preF[i]; }
    }
    __device__ __forceinline__ void storeF(int k)
    {
        double* dst = W + wo.F + (long)k * FR;
        const int nu_ = bnd(k) ? WK::f_cf(MNU) : WK::f_cf(MMID);   // (the row coefficients behind it are written by factor_pre)
        for (int idx = lane; idx < nu_; idx += 64) dst[idx] = L->st.F[idx];
    }
    // views of the staged factor record; mm = nu-rows of the node the record belongs to (leading dimension of Lni, Y)
    __device__ __forceinline__ double* Li() const { return L->st.F; }
    __device__ __forceinline__ double* Lni(int mm) const { return L->st.F + WK::f_lni(mm); }
    __device__ __forceinline__ double* Xm(int mm) const { return L->st.F + WK::f_x(mm); }
    __device__ __forceinline__ double* Ym(int mm) const { return L->st.F + WK::f_y(mm); }
    __device__ __forceinline__ double* Cf(int mm) const { return L->st.F + WK::f_cf(mm); }
    // row-record / cone-scaling / primal prefetch (one node ahead), committed to LDS at the top of the node
    __device__ __forceinline__ void pf_rows(double (&r)[NROWR], const double* v, int k) const
    {
#pragma unroll
        for (int i = 0; i < NROWR; i++) { const int idx = lane + 64 * i; r[i] = v[(long)k * RS + (idx < RS ? idx : RS - 1)]; }
    }
    __device__ __forceinline__ void cm_rows(double* dst, const double (&r)[NROWR]) const
    {
#pragma unroll
        for (int i = 0; i < NROWR; i++) { const int idx = lane + 64 * i; if (idx < RS) dst[idx] = r[i]; }
    }
    // cone scalings for the horizon sweeps: only W^-1 (entries [16, 32) of each 36-double cone record) is read there
    static constexpr int NSOCW = (NSOC1 * 16 + 63) / 64;
    __device__ __forceinline__ void pf_soc(int k)
    {
        const double* src = W + wo.socW + (long)k * nsoc * 36;
#pragma unroll
        for (int i = 0; i < NSOCW; i++) {
            int idx = lane + 64 * i;
            idx = idx < nsoc * 16 ? idx : (nsoc > 0 ? nsoc * 16 - 1 : 0);
            pS[i] = src[(idx / 16) * 36 + 16 + idx % 16];
        }
    }
    __device__ __forceinline__ void cm_soc()
    {
#pragma unroll
        for (int i = 0; i < NSOCW; i++) { const int idx = lane + 64 * i; if (idx < nsoc * 16) L->soc[(idx / 16) * 36 + 16 + idx % 16] = pS[i]; }
    }
    __device__ __forceinline__ void load_rows(double* dst, const double* v, int k) const
    {
        for (int r = lane; r < RS; r += 64) dst[r] = v[(long)k * RS + r];
    }
    __device__ __forceinline__ void load_grows(double* dst, const double* v) const
    {
        for (int r = lane; r < RG; r += 64) dst[r] = v[(long)N * RS + r];
    }
    // stage-record field views (LDS)
    __device__ __forceinline__ const double* D() const { return L->st.Pk + S::O_D; }
    __device__ __forceinline__ const double* E() const { return L->st.Pk + S::O_E; }
    __device__ __forceinline__ const double* Fp() const { return L->st.Pk + S::O_FP; }
    __device__ __forceinline__ const double* Kl() const { return L->st.Pk + S::O_KL; }
    __device__ __forceinline__ const double* Kp() const { return L->st.Pk + S::O_KP; }
    // global-record views
    __device__ __forceinline__ const double* gH0() const { return L->G + S::Q_H0; }
    __device__ __forceinline__ const double* gK0() const { return L->G + S::Q_K0; }
    __device__ __forceinline__ const double* gHf() const { return L->G + S::Q_HF; }
    __device__ __forceinline__ const double* gKf() const { return L->G + S::Q_KF; }
    __device__ __forceinline__ const double* gLp() const { return L->G + S::Q_LP; }

    // ---------------- out = G * v (linear part of the rows) ----------------
    // Round 6: FLAT over the rows, by row type (before: one node at a time, the node's records staged through LDS -- 750
    // instructions and one exposed memory round trip per node).  A lane owns one (node, row) item, reads that row's coefficients
    // straight from the slab (the L1 absorbs the stride), items are processed two at a time with all loads ahead of the stores;
    // lanes past the end repeat the last item.
    template <class F>
    __device__ __forceinline__ void flat_items(int count, F&& f) const
    {
#pragma unroll 1
        for (int base = 0; base < count; base += 128) {
            const int i0 = base + lane, i1 = base + 64 + lane;
            auto r0 = f(i0 < count ? i0 : count - 1), r1 = f(i1 < count ? i1 : count - 1);
            r0.store(); r1.store();
        }
    }
    __device__ __forceinline__ void G_apply(double* v, double* out)
    {
        const long long t0_ = tick();
        double pv_[npa];
#pragma unroll
        for (int j = 0; j < npa; j++) pv_[j] = np > 0 ? PV(v, j) : 0.0;
        struct R2 { double* p0; double* p1; double v0, v1; __device__ __forceinline__ void store() const { *p0 = v0; *p1 = v1; } };
        // dynamics rows: +-(D z_k + E z_{k+1} + Fp p) - y   (absent at the last node: zeros)
        flat_items(N * nx, [&](int idx) {
            const int k = idx / nx, i = idx - k * nx, kn = k + 1 < N ? k + 1 : N - 1;
            const double* Pk_ = Pg + (long)k * SR;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < nz; j++) acc += Pk_[S::O_D + i * nz + j] * Z(v, k, j) + Pk_[S::O_E + i * nz + j] * Z(v, kn, j);
#pragma unroll
            for (int j = 0; j < np; j++) acc += Pk_[S::O_FP + i * npa + j] * pv_[j];
            const double y = AUX(v, k, S::A_Y + i);
            const bool lastk = k == N - 1;
            return R2{&ROW(out, k, i), &ROW(out, k, nx + i), lastk ? 0.0 : acc - y, lastk ? 0.0 : -acc - y};
        });
        // local rows: hinge (+ its pair), linear, cone
        if (ml > 0) {
            flat_items(N * ml, [&](int idx) {
                const int k = idx / ml, row = idx - k * ml;
                const double* Pk_ = Pg + (long)k * SR;
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < nz; j++) acc += Pk_[S::O_KL + row * nz + j] * Z(v, k, j);
#pragma unroll
                for (int j = 0; j < np; j++) acc += Pk_[S::O_KP + row * npa + j] * pv_[j];
                const bool hg = row < ns, lin = row < ns + nl;
                const double vv = AUX(v, k, S::A_V + (hg ? row : 0));
                const int r = hg ? S::R_H0 + row : (lin ? S::R_LIN + row - ns : S::R_SOC + row - ns - nl);
                const double val = hg ? acc - vv : (lin ? acc : -acc);
                // second store: the hinge row's partner (-v); other kinds write the same value to the same slot twice
                return R2{&ROW(out, k, r), &ROW(out, k, hg ? S::R_H1 + row : r), val, hg ? -vv : val};
            });
        }
        // trust-region rows: +-z_j - eta
        flat_items(N * nz, [&](int idx) {
            const int k = idx / nz, j = idx - k * nz;
            const double zj = Z(v, k, j), eta = AUX(v, k, j < nx ? S::A_EX : S::A_EU);
            return R2{&ROW(out, k, S::R_TR0 + j), &ROW(out, k, S::R_TR1 + j), zj - eta, -zj - eta};
        });
        // global rows
        for (int r = lane; r < 2 * nic; r += 64) {
            const int i = r % nic;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < nx; j++) acc += gH0()[i * nx + j] * Z(v, 0, j);
#pragma unroll
            for (int j = 0; j < np; j++) acc += gK0()[i * npa + j] * pv_[j];
            GROW(out, r) = (r < nic ? acc : -acc) - GAUX(v, S::GA_YIC + i);
        }
        for (int r = lane; r < 2 * ntc; r += 64) {
            const int i = r % ntc;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < nx; j++) acc += gHf()[i * nx + j] * Z(v, N - 1, j);
#pragma unroll
            for (int j = 0; j < np; j++) acc += gKf()[i * npa + j] * pv_[j];
            GROW(out, S::G_TC0 + r) = (r < ntc ? acc : -acc) - GAUX(v, S::GA_YTC + i);
        }
        for (int r = S::G_TRP0 + lane; r < RG; r += 64) {
            double val;
            if (r < S::G_LIN) {
                const int j = (r - S::G_TRP0) % (np > 0 ? np : 1);
                val = (r < S::G_TRP1 ? PV(v, j) : -PV(v, j)) - GAUX(v, S::GA_EP);
            } else {
                const int i = r - S::G_LIN;
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < np; j++) acc += gLp()[i * npa + j] * pv_[j];
                val = acc;
            }
            GROW(out, r) = val;
        }
        gsync();
        PROF_ADD(0, tick() - t0_);
    }
    // main-variable part of row r of the staged node (uses Pk, zk, zn, pv)
    __device__ __forceinline__ double row_main(int k, int r) const
    {
        if (r < 2 * nx) {
            if (k >= N - 1) return 0.0;
            const int i = r % nx;
            double acc = 0.0;
            const double *d = D() + i * nz, *e = E() + i * nz;
#pragma unroll
            for (int j = 0; j < nz; j++) acc += d[j] * L->zk[j] + e[j] * L->zn[j];
#pragma unroll
            for (int j = 0; j < np; j++) acc += Fp()[i * npa + j] * L->pv[j];
            return r < nx ? acc : -acc;
        }
        if (r < S::R_TR0) {
            if (r >= S::R_H1) return 0.0;
            const int i = r - S::R_H0;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < nz; j++) acc += Kl()[i * nz + j] * L->zk[j];
#pragma unroll
            for (int j = 0; j < np; j++) acc += Kp()[i * npa + j] * L->pv[j];
            return acc;
        }
        if (r < S::R_LIN) {
            const int j = (r - S::R_TR0) % nz;
            return r < S::R_TR1 ? L->zk[j] : -L->zk[j];
        }
        const int row = ns + (r - S::R_LIN);
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < nz; j++) acc += Kl()[row * nz + j] * L->zk[j];
#pragma unroll
        for (int j = 0; j < np; j++) acc += Kp()[row * npa + j] * L->pv[j];
        return r < S::R_SOC ? acc : -acc;
    }
    // aux-variable part of row r: value subtracted (uses ak)
    __device__ __forceinline__ double row_aux(int r) const
    {
        if (r < 2 * nx) return L->ak[S::A_Y + r % nx];
        if (r < S::R_TR0) return L->ak[S::A_V + (r - S::R_H0) % (ns > 0 ? ns : 1)];
        if (r < S::R_LIN) return L->ak[((r - S::R_TR0) % nz) < nx ? S::A_EX : S::A_EU];
        return 0.0;
    }

    // ---------------- out = G' * mu ----------------
    // flat like G_apply: items (node, component) for the z part, (node, aux) for the epigraph part, (node, row) for the p part.
    // Written BRANCH-FREE (clamped indices, 0 / 1 factors): a select between two loads is compiled into a branch around each load,
    // and every such branch is a separate exposed memory round trip (measured: 2.2 x slower than the node loop it replaced).
    __device__ __forceinline__ void GT_apply(double* mu, double* out)
    {
        const long long t0_ = tick();
        struct R1 { double* p0; double v0; __device__ __forceinline__ void store() const { *p0 = v0; } };
        const long NRS = (long)N * RS;
        flat_items(N * nz, [&](int idx) {
            const int k = idx / nz, j = idx - k * nz, kp = k > 0 ? k - 1 : 0, jx = j < nx ? j : nx - 1;
            const double* Pk_ = Pg + (long)k * SR;
            const double* Pp_ = Pg + (long)kp * SR;
            const double* mk = mu + (long)k * RS;
            const double* mp = mu + (long)kp * RS;
            const double fl = k == N - 1 ? 0.0 : 1.0, ff = k == 0 ? 0.0 : 1.0;
            const double bic = (k == 0 && j < nx) ? 1.0 : 0.0, btc = (k == N - 1 && j < nx) ? 1.0 : 0.0;
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < nx; i++) acc += Pk_[S::O_D + i * nz + j] * (fl * (mk[i] - mk[nx + i]));
#pragma unroll
            for (int i = 0; i < nx; i++) acc += Pp_[S::O_E + i * nz + j] * (ff * (mp[i] - mp[nx + i]));
            acc += mk[S::R_TR0 + j] - mk[S::R_TR1 + j];
#pragma unroll
            for (int i = 0; i < ns; i++) acc += Pk_[S::O_KL + i * nz + j] * mk[S::R_H0 + i];
#pragma unroll
            for (int i = 0; i < nl; i++) acc += Pk_[S::O_KL + (ns + i) * nz + j] * mk[S::R_LIN + i];
#pragma unroll
            for (int i = 0; i < 4 * nsoc; i++) acc -= Pk_[S::O_KL + (ns + nl + i) * nz + j] * mk[S::R_SOC + i];
            double aic = 0.0, atc = 0.0;
#pragma unroll
            for (int i = 0; i < nic; i++) aic += gH0()[i * nx + jx] * (mu[NRS + S::G_IC0 + i] - mu[NRS + S::G_IC1 + i]);
#pragma unroll
            for (int i = 0; i < ntc; i++) atc += gHf()[i * nx + jx] * (mu[NRS + S::G_TC0 + i] - mu[NRS + S::G_TC1 + i]);
            acc += bic * aic;
            acc += btc * atc;
            return R1{&Z(out, k, j), acc};
        });
        // epigraph part: y (dynamics pairs) and v (hinge pairs)
        flat_items(N * (nx + ns), [&](int idx) {
            const int k = idx / (nx + ns), i = idx - k * (nx + ns);
            const bool isd = i < nx;
            const int ra = isd ? i : S::R_H0 + (i - nx), rb = isd ? nx + i : S::R_H1 + (i - nx);
            const double f = (isd && k == N - 1) ? 0.0 : -1.0;
            return R1{&AUX(out, k, i), f * (ROW(mu, k, ra) + ROW(mu, k, rb))};
        });
        // ... eta_x, eta_u (trust-region pairs of the group)
        flat_items(N * 2, [&](int idx) {
            const int k = idx >> 1, which = idx & 1;
            constexpr int NM = nx > nu ? nx : nu;
            const int j0 = which == 0 ? 0 : nx, n = which == 0 ? nx : nu;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < NM; q++) {
                const int jq = j0 + (q < n ? q : n - 1);
                acc -= (q < n ? 1.0 : 0.0) * (ROW(mu, k, S::R_TR0 + jq) + ROW(mu, k, S::R_TR1 + jq));
            }
            return R1{&AUX(out, k, which == 0 ? S::A_EX : S::A_EU), acc};
        });
        if (np > 0) {
            double pacc[npa];
#pragma unroll
            for (int j = 0; j < npa; j++) pacc[j] = 0.0;
            constexpr int NR = nx + ml;
            const int cnt = N * NR;
#pragma unroll 1
            for (int base = 0; base < cnt; base += 128) {
                double m_[2], c_[2][npa];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int i0 = base + 64 * u + lane, idx = i0 < cnt ? i0 : cnt - 1;
                    const int k = idx / NR, r = idx - k * NR;
                    const double* Pk_ = Pg + (long)k * SR;
                    const int row = r - nx;
                    const bool isd = r < nx, hg = row < ns, lin = row < ns + nl;
                    const int ra = isd ? r : (hg ? S::R_H0 + row : (lin ? S::R_LIN + row - ns : S::R_SOC + row - ns - nl));
                    const int rb = isd ? nx + r : ra;
                    const double fa = (i0 < cnt) ? ((isd && k == N - 1) ? 0.0 : ((isd || lin) ? 1.0 : -1.0)) : 0.0;   // cone rows enter with -1
                    const double fb_ = isd ? 1.0 : 0.0;
                    m_[u] = fa * (ROW(mu, k, ra) - fb_ * ROW(mu, k, rb));
                    const int ci = isd ? S::O_FP + r * npa : S::O_KP + row * npa;
#pragma unroll
                    for (int j = 0; j < np; j++) c_[u][j] = Pk_[ci + j];
                }
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int j = 0; j < np; j++) pacc[j] += c_[u][j] * m_[u];
            }
#pragma unroll
            for (int j = 0; j < np; j++) {
                double t = wave_sum(pacc[j]);
                if (lane == 0) {
                    t += GROW(mu, S::G_TRP0 + j) - GROW(mu, S::G_TRP1 + j);
                    for (int i = 0; i < ng; i++) t += gLp()[i * npa + j] * GROW(mu, S::G_LIN + i);
                    for (int i = 0; i < nic; i++) t += gK0()[i * npa + j] * (GROW(mu, S::G_IC0 + i) - GROW(mu, S::G_IC1 + i));
                    for (int i = 0; i < ntc; i++) t += gKf()[i * npa + j] * (GROW(mu, S::G_TC0 + i) - GROW(mu, S::G_TC1 + i));
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
        gsync();
        PROF_ADD(1, tick() - t0_);
    }

    // ---------------- constants: hneg (= -h), cost vector cv, diagonal qd on the xi layout ----------------
    __device__ __forceinline__ void build_constants(double* hn, double* cv, double* qd)
    {
        prefetch(0);
        for (int k = 0; k < N; k++) {
            commit();
            sync();
            if (k + 1 < N) prefetch(k + 1);
            for (int r = lane; r < RS; r += 64) {
                double c = 0.0;
                if (r < 2 * nx) { if (k < N - 1) { const double v = L->st.Pk[S::O_CD + r % nx]; c = r < nx ? v : -v; } }
                else if (r < S::R_H1) c = L->st.Pk[S::O_CL + (r - S::R_H0)];
                else if (r < S::R_TR0) c = 0.0;
                else if (r < S::R_LIN) { const int j = (r - S::R_TR0) % nz; const double v = L->st.Pk[S::O_ZREF + j]; c = r < S::R_TR1 ? -v : v; }
                else if (r < S::R_SOC) c = L->st.Pk[S::O_CL + ns + (r - S::R_LIN)];
                else c = -L->st.Pk[S::O_CL + ns + nl + (r - S::R_SOC)];
                ROW(hn, k, r) = c;
            }
            if (lane < nz) { Z(cv, k, lane) = L->st.Pk[S::O_Q + lane]; Z(qd, k, lane) = L->st.Pk[S::O_QD + lane]; }
            else if (lane < nz + AS) {
                const int i = lane - nz;
                double c;
                if (i < nx) c = k < N - 1 ? L->st.Pk[S::O_OM + i] : 0.0;
                else if (i < nx + ns) c = L->st.Pk[S::O_HW + i - nx];
                else c = L->st.Pk[S::O_TTR];
                AUX(cv, k, i) = c; AUX(qd, k, i) = 0.0;
            }
            sync();
        }
        for (int r = lane; r < RG; r += 64) {
            double c;
            if (r < S::G_TC0) { const double v = L->G[S::Q_L0 + r % nic]; c = r < S::G_IC1 ? v : -v; }
            else if (r < S::G_TRP0) { const double v = L->G[S::Q_LF + (r - S::G_TC0) % ntc]; c = r < S::G_TC1 ? v : -v; }
            else if (r < S::G_LIN) { const double v = L->G[S::Q_PREF + (r - S::G_TRP0) % (np > 0 ? np : 1)]; c = r < S::G_TRP1 ? -v : v; }
            else c = L->G[S::Q_LPC + r - S::G_LIN];
            GROW(hn, r) = c;
        }
        if (lane < npa) { PV(cv, lane) = np > 0 ? L->G[S::Q_QPL + lane] : 0.0; PV(qd, lane) = np > 0 ? L->G[S::Q_QP + lane] : 0.0; }
        for (int i = lane; i < AG; i += 64) {
            double c;
            if (i < nic) c = L->G[S::Q_BW0 + i];
            else if (i < nic + ntc) c = L->G[S::Q_BWF + i - nic];
            else c = np > 0 ? ttrp : 0.0;
            GAUX(cv, i) = c; GAUX(qd, i) = 0.0;
        }
        gsync();
    }

    // ---------------- small dense helpers on LDS ----------------
    // in-place lower Cholesky of the n x n row-major matrix A (ld), n compile time
    template <int n, int ld>
    __device__ __forceinline__ void chol(double* A) const
    {
#pragma unroll 1
        for (int j = 0; j < n; j++) {
            const double d = A[j * ld + j];
            if (lane == 0 && !(d > 0.0)) L->fail = 1;
            const double djj = sqrt(d > 0.0 ? d : 1.0);
            sync();
            if (lane == 0) A[j * ld + j] = djj;
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
    // Linv = inverse of the lower-triangular n x n matrix Lm (column c handled by lane c)
    template <int n, int ld>
    __device__ __forceinline__ void tri_inverse(const double* Lm, double* Linv) const
    {
        for (int c = lane; c < n; c += 64) {
#pragma unroll 1
            for (int i = 0; i < n; i++) {
                double acc = (i == c) ? 1.0 : 0.0;
                for (int j = c; j < i; j++) acc -= Lm[i * ld + j] * Linv[j * ld + c];
                Linv[i * ld + c] = (i < c) ? 0.0 : acc / Lm[i * ld + i];
            }
        }
        sync();
    }

    struct Pair {
        double kap, tau, rth, Wt;
    };
    __device__ __forceinline__ static Pair pairA(double w1, double w2, double rt1, double rt2, double rxa)
    {
        Pair p;
        const double r1 = w1 * rt1, r2 = w2 * rt2;
        p.Wt = w1 + w2; p.rth = -rxa + r1 + r2;
        p.kap = 4.0 * w1 * w2 / p.Wt;
        p.tau = -(r1 - r2) + (w1 - w2) * p.rth / p.Wt;
        return p;
    }
    __device__ __forceinline__ static Pair pairC(double w1, double w2, double rt1, double rt2, double rxa)
    {
        Pair p;
        const double r1 = w1 * rt1, r2 = w2 * rt2;
        p.Wt = w1 + w2; p.rth = -rxa + r1 + r2;
        p.kap = w1 * w2 / p.Wt;
        p.tau = -r1 + w1 * p.rth / p.Wt;
        return p;
    }
    __device__ __forceinline__ bool nu_live(int k, int c) const
    {
        if (c < nx) return k < N - 1;
        if (c < nx + ns) return true;
        const int i = c - nx - ns;
        return (k == 0 && i < nic) || (k == N - 1 && i < ntc);
    }
    // variants for code instantiated for MID nodes only (MM == MNU_MID < MNU): every nu-row c < MM is a live dynamics
    // or hinge row there, so the node-type predicates and the boundary-condition branches fold away
    template <bool MID>
    __device__ __forceinline__ bool nu_live_m(int k, int c) const
    {
        if constexpr (MID) return true; else return nu_live(k, c);
    }
    template <bool MID>
    __device__ __forceinline__ double Dt_m(int k, int c, int j) const
    {
        if constexpr (MID) return c < nx ? D()[c * nz + j] : Kl()[(c - nx) * nz + j];
        else return Dt(k, c, j);
    }
    template <bool MID>
    __device__ __forceinline__ double Ft_m(int k, int c, int j) const
    {
        if constexpr (MID) return c < nx ? Fp()[c * npa + j] : Kp()[(c - nx) * npa + j];
        else return Ft(k, c, j);
    }
    // Dt_k[c][j], Ft_k[c][j] from the staged records
    __device__ __forceinline__ double Dt(int k, int c, int j) const
    {
        if (c < nx) return D()[c * nz + j];           // zero record at the last node
        if (c < nx + ns) return Kl()[(c - nx) * nz + j];
        const int i = c - nx - ns;
        if (j >= nx) return 0.0;
        if (k == 0 && i < nic) return gH0()[i * nx + j];
        if (k == N - 1 && i < ntc) return gHf()[i * nx + j];
        return 0.0;
    }
    __device__ __forceinline__ double Ft(int k, int c, int j) const
    {
        if (c < nx) return Fp()[c * npa + j];
        if (c < nx + ns) return Kp()[(c - nx) * npa + j];
        const int i = c - nx - ns;
        if (k == 0 && i < nic) return gK0()[i * npa + j];
        if (k == N - 1 && i < ntc) return gKf()[i * npa + j];
        return 0.0;
    }
    // (w1, w2, rtil1, rtil2, rxaux) of nu-row c of node k from staged rows: wr (weights), rt (rtil), rxa (aux part of rx)
    __device__ __forceinline__ void nu_row_data(int k, int c, const double* wr, const double* rt, const double* wg, const double* rtg,
                                const double* rxak, const double* rxga, double& w1, double& w2, double& t1, double& t2,
                                double& rxa, bool& hinge) const
    {
        hinge = false;
        if (c < nx) { w1 = wr[c]; w2 = wr[nx + c]; t1 = rt[c]; t2 = rt[nx + c]; rxa = rxak[S::A_Y + c]; }
        else if (c < nx + ns) { const int i = c - nx; hinge = true; w1 = wr[S::R_H0 + i]; w2 = wr[S::R_H1 + i]; t1 = rt[S::R_H0 + i]; t2 = rt[S::R_H1 + i]; rxa = rxak[S::A_V + i]; }
        else {
            const int i = c - nx - ns;
            if (k == 0) { w1 = wg[S::G_IC0 + i]; w2 = wg[S::G_IC1 + i]; t1 = rtg[S::G_IC0 + i]; t2 = rtg[S::G_IC1 + i]; rxa = rxga[S::GA_YIC + i]; }
            else { w1 = wg[S::G_TC0 + i]; w2 = wg[S::G_TC1 + i]; t1 = rtg[S::G_TC0 + i]; t2 = rtg[S::G_TC1 + i]; rxa = rxga[S::GA_YTC + i]; }
        }
    }
    // type-B (L_inf block) Schur complement entry (a,b): weights w1[j], w2[j] (LDS), j in [0,n)
    template <int n>
    __device__ __forceinline__ static double typeB_entry(const double* w1, const double* w2, int a_, int b_)
    {
        double Wt = 0.0;
#pragma unroll
        for (int j = 0; j < n; j++) Wt += w1[j] + w2[j];
        const double ha = w1[a_] - w2[a_];
        if (a_ != b_) return -ha * (w1[b_] - w2[b_]) / Wt;
        double rest = 0.0;
#pragma unroll
        for (int j = 0; j < n; j++) rest += (j != a_) ? (w1[j] + w2[j]) : 0.0;
        const double d = w1[a_] + w2[a_];
        return 4.0 * w1[a_] * w2[a_] / d + ha * ha * rest / (d * Wt);
    }

    __device__ __forceinline__ void factor_pre(const double* w, double (&Dp)[npa * npa]);
    template <int MM, int MP>
    __device__ __forceinline__ void factor_stage(int k, const double (&pH)[4], double pCf, double pC0);
    template <bool FWD, int MM>
    __device__ __forceinline__ int pad_off(int p) const;
    template <bool FWD>
    __device__ __forceinline__ void pad_begin(int (&offM)[NPREF_MID], int (&offB)[NPREF]);
    template <int NO, int NV>
    __device__ __forceinline__ void commit_pad(const int (&off)[NO], const double (&v)[NV], int i0);
#ifndef SCP_K3_PD
#define SCP_K3_PD 3
#endif
    static constexpr int PD = SCP_K3_PD;   // prefetch distance (nodes) of the chain sweeps
    __device__ __forceinline__ void chain_load(int k, double (&f)[NPREF_MID], double& b, double& t, const double* bv, const double* tv, int lz_, int lm_) const;
    __device__ __forceinline__ void chain_load_bnd(int k, double (&f)[NPREF], double& b, double& t, const double* bv, const double* tv, int lz_, int lm_) const;
    template <int MM>
    __device__ __forceinline__ double fwd_chain(int k, double znx);
    template <int MM>
    __device__ __forceinline__ double bwd_chain(int k, double zn, double* zo, double* nuo);
    __device__ __forceinline__ void bwd_sweep(const double* bh_v, const double* th_v, double* zo, double* nuo);
    template <int NC>
    __device__ __forceinline__ void arrow_dot(const double* yz, const double* yn, double (&acc)[npa * NC]);
    __device__ __forceinline__ void newton_rhs(const double* w, const double* rtil, const double* rxv, double (&bp)[npa]);
    __device__ __forceinline__ void factor(double* w);
    __device__ __forceinline__ void solve_backward_cols();
    __device__ __forceinline__ void newton_solve(double* w, double* rtil, double* rxv, double* dxi);
    __device__ __forceinline__ void finish_direction(double* w, double* rtil, double* rxv, double* dxi, double* gd, double* dl);
    __device__ __forceinline__ void nt_update(double* s, double* lam);
    __device__ __forceinline__ void nt_identity();
    __device__ __forceinline__ static double soc_step(const double* s, const double* d);
    __device__ __forceinline__ double min_margin(double* v, double* dv, double alpha) const;
    template <int WPE>
    __device__ __forceinline__ void run();
};

}  // namespace scp

#include "ipm2_newton.hpp"
#include "ipm2_run.hpp"
