// K1: `discretize!` (FOH) as a hand-written gfx950 kernel.
//
// Reference semantics: src/solvers/discretization.jl:160-217 (loop over the N-1
// intervals), :235-286 (derivs_foh), :354-406 (set_update_matrices) and the
// classic RK4 of src/utils/helper.jl:411-424,451-501 on the sub-grid
// LinRange(t_k, t_{k+1}, Nsub).  The formulation is the reference's own
// (integrate V = [x; Phi; int Phi^-1 B-; int Phi^-1 B+; int Phi^-1 F;
// int Phi^-1 r; int Phi^-1 E], multiply by Phi at the end) so results agree
// with the CPU oracle to fp64 round-off, not merely to RK4 truncation error.
//
// MI355X mapping (wave64):
//   * one LANE GROUP (32 or 64 lanes) per (problem b, interval k); the group's
//     lanes each own ONE COLUMN of the augmented matrix
//         [ Phi | V_B- | V_B+ | V_F | V_r | V_E ]      (nx rows, NCOL columns)
//     in registers, for all of RK4 (state, accumulator, stage value).  The
//     state x is carried redundantly by every lane (nx doubles), so the model
//     f/A/B/F is evaluated without any cross-lane traffic.
//   * Phi^-1 * [s-B, s+B, F, r, E] is ONE cooperative Gaussian elimination with
//     partial pivoting on the augmented matrix [Phi | rhs]: lane s of the group
//     finds the pivot and broadcasts the multipliers with ds_bpermute
//     (__shfl); every lane updates its own column; back-substitution
//     broadcasts U's entries the same way.  nx^2 FMAs + nx^2 broadcasts per
//     lane per stage, no LDS allocation, no barriers.
//   * d(Phi)/dt = A*Phi is a private mat-vec per Phi-lane; model Jacobians are
//     built in registers with compile-time zeros (fully unrolled), so the
//     compiler drops structurally-zero products.
//   * outputs are written once, column-per-lane, contiguous across the lanes of
//     a group (A, B-, B+, F, E blocks are adjacent columns of the same matrix).
//
// The kernel is fp64 VALU-latency bound: algorithmic intensity
// is 30-1000 flop/B, far right of the MI355X fp64 ridge (~10 flop/B).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "models/model_common.hpp"

namespace scp {

struct DiscArgs {
    int B, N, Nsub;
    const double* xd;   // [nx,N,B]
    const double* ud;   // [nu,N,B]
    const double* p;    // [np,B]
    const double* iSx;  // [nx] diag of inv(Sx)
    double feas_tol;
    double* A;       // [nx,nx,N-1,B]
    double* Bm;      // [nx,nu,N-1,B]
    double* Bp;      // [nx,nu,N-1,B]
    double* F;       // [nx,npF,N-1,B]
    double* r;       // [nx,N-1,B]
    double* E;       // [nx,nx,N-1,B]
    double* defect;  // [nx,N-1,B]
    int* feas;       // [B], pre-set to 1; AND-reduced with atomicAnd
    const int* mask; // optional [B]: problems with mask[b] == 0 are skipped (converged in the SCP loop)
};

// Julia LinRange(a,b,n)[j] (0-based j), Base `lerpi`: (1-j/(n-1))*a + (j/(n-1))*b
__device__ __forceinline__ double linrange(double a, double b, int n, int j)
{
    const double tt = (double)j / (double)(n - 1);
    return (1.0 - tt) * a + tt * b;
}

enum Role { R_PHI = 0, R_BM = 1, R_BP = 2, R_F = 3, R_R = 4, R_E = 5, R_IDLE = 6 };

// 1 / a for the pivots of the cooperative LU.  Default: the IEEE division.  -DSCP_K1_RCP: hardware estimate + two Newton steps (7
// instructions instead of ~28; 2 (nx - lead) of them per stage: free-flyer 208 -> 201 ms, Starship 23.3 -> 22.5 ms, discretize! parity 1e-10 /
// 1e-8 unchanged) -- measured in round 6 and NOT shipped: the last-bit differences move three loop-level parity tests of the generic path
// (Starship / free-flyer SCvx loops against the oracle's records fork at a degenerate LP: tests/test_starship_gpu.py, test_freeflyer_gpu.py;
// gpurun_out/r06_k1ab), and 3 % of K1 does not buy that.
template <class T>
__device__ __forceinline__ T pivot_rcp(T a)
{
#ifdef SCP_K1_RCP
    if constexpr (sizeof(T) == 8) {
        double y = __builtin_amdgcn_rcp((double)a);
        y = y * (2.0 - (double)a * y);
        y = y * (2.0 - (double)a * y);
        return (T)y;
    } else
#endif
        return (T)1 / a;
}

// Broadcast of lane (gbase + src) of a lane group to the group; src is a compile-time constant after unrolling.  A 64-lane group is
// the whole wavefront: v_readlane_b32 into scalar registers -- one VALU pass, the value then feeds the FMAs as a scalar operand --
// instead of ds_bpermute_b32 through the LDS crossbar and an lgkmcnt wait (free-flyer: 1 738 ds_bpermute, 790 waits and 153 scratch
// accesses per RK4 step before; 0 / 35 / 0 after; reference-form launch 562 -> 433 ms, 344 ms with the scalar-branch row exchange below).  Two 32-lane groups per wavefront have two
// sources: reading both and selecting per half was measured SLOWER than the shuffle (Starship 28.4 -> 37.4 ms), so they keep it.
template <int G, class V>
__device__ __forceinline__ V group_bcast(V v, int gbase, int src)
{
    static_assert(G == 32 || G == 64, "lane groups are half or whole wavefronts");
    if constexpr (G == 32) return __shfl(v, gbase + src);
    else if constexpr (sizeof(V) == 8)
        return (V)__hiloint2double(__builtin_amdgcn_readlane(__double2hiint((double)v), src), __builtin_amdgcn_readlane(__double2loint((double)v), src));
    else if constexpr (sizeof(V) == 4 && !std::is_integral<V>::value)
        return (V)__int_as_float(__builtin_amdgcn_readlane(__float_as_int((float)v), src));
    else return (V)__builtin_amdgcn_readlane((int)v, src);
}

template <class M>
struct DiscLayout {
    static constexpr int nx = M::nx, nu = M::nu, npF = M::npF;
    static constexpr int NCOL = 2 * nx + 2 * nu + npF + 1;
    static constexpr int G = NCOL <= 32 ? 32 : 64;  // lanes per (problem, interval)
    static_assert(NCOL <= 64, "augmented matrix wider than a wavefront");
    static constexpr int GROUPS_PER_BLOCK = 256 / G;
};

// T: the arithmetic type of the whole integration (state, Phi, the LU, the RK4 accumulators, the model evaluation).
// T = double is the reference's arithmetic (basic_types.jl:31); T = float is the "fp64 vs fp32 tolerance check" variant
// of BASELINE.json configs[2] (inputs / outputs stay fp64 arrays; scp_set_discretize_precision).
// IMP = true: the IMPULSE discretisation (:186-193, derivs_impulse :304-340, :384-390) on the same lane layout: the
// state starts from x_k + f(t_k, -k, x_k, u_k, p) (the model's impulse response, M::impulse), the input is zero between
// the nodes, the B-/B+ lanes integrate nothing and B- finally receives A_k * B(t_k, -k, x_k, u_k, p); B+ is zero
// (the reference's DLTV holds a single B for IMPULSE, discretization.jl:31,60-66).
template <class M, bool IMP = false, class T = double>
__global__ __launch_bounds__(256) void discretize_foh_kernel(DiscArgs a, typename M::Params par)
{
    using L = DiscLayout<M>;
    constexpr int nx = M::nx, nu = M::nu, npF = M::npF;
    constexpr int npFa = npF > 0 ? npF : 1;
    constexpr int G = L::G;

    const int lane = threadIdx.x & 63;
    const int gl = lane % G;        // lane within the group == column index
    const int gbase = lane - gl;    // first lane of the group inside the wave
    const long total = (long)a.B * (a.N - 1);
    const long gid_raw = ((long)blockIdx.x * 256 + threadIdx.x) / G;
    const bool active = gid_raw < total;
    const long gid = active ? gid_raw : total - 1;  // keep shuffles well-defined
    const int b = (int)(gid / (a.N - 1));
    const int k = (int)(gid % (a.N - 1));  // 0-based interval; reference k = k+1
    const bool write = active && (a.mask == nullptr || a.mask[b] != 0);
    if (!write) return;   // the whole group leaves together: every shuffle below has its source lane inside the group

    // ---- role of this lane (which column of the augmented matrix it owns) ----
    int role, ridx;
    if (gl < nx) { role = R_PHI; ridx = gl; }
    else if (gl < nx + nu) { role = R_BM; ridx = gl - nx; }
    else if (gl < nx + 2 * nu) { role = R_BP; ridx = gl - nx - nu; }
    else if (gl < nx + 2 * nu + npF) { role = R_F; ridx = gl - nx - 2 * nu; }
    else if (gl < nx + 2 * nu + npF + 1) { role = R_R; ridx = 0; }
    else if (gl < L::NCOL) { role = R_E; ridx = gl - (nx + 2 * nu + npF + 1); }
    else { role = R_IDLE; ridx = 0; }

    // ---- inputs (every lane of the group loads the same few words: L1 broadcast) ----
    const double* xk = a.xd + ((long)b * a.N + k) * nx;
    const double* uk = a.ud + ((long)b * a.N + k) * nu;
    const double* pb = a.p + (long)b * np_total<M>(a.N);   // p = [global; node parameters]: the dynamics read the globals
    T x[nx], u0[nu], u1[nu], pF[npFa];
#pragma unroll
    for (int i = 0; i < nx; i++) x[i] = xk[i];  // V0[x] = xd[:,k]  (:185)
#pragma unroll
    for (int i = 0; i < nu; i++) { u0[i] = uk[i]; u1[i] = uk[nu + i]; }
#pragma unroll
    for (int j = 0; j < npFa; j++) pF[j] = (npF > 0) ? pb[M::Fcol(j)] : 0.0;

    const T t0 = (T)linrange(0.0, 1.0, a.N, k);      // t_grid = LinRange(0,1,N), scp.jl:147
    const T t1 = (T)linrange(0.0, 1.0, a.N, k + 1);
    T bimp[nx];   // IMPULSE: this lane's column of B(t_k, -k, x_k, u_k, p) (B- lanes)
#pragma unroll
    for (int i = 0; i < nx; i++) bimp[i] = 0.0;
    if constexpr (IMP) {
        T dx[nx], Bi[nx * nu];
        M::impulse(par, t0, k + 1, x, u0, pb, dx, Bi);      // f(t_k, -k, ...) and B(t_k, -k, ...)
#pragma unroll
        for (int i = 0; i < nx; i++) x[i] += dx[i];          // xk_plus = xk + f(tk, -k, xk, uk, p)  (:191-192)
        if (role == R_BM) {
#pragma unroll
            for (int i = 0; i < nx; i++) {
                T v = 0.0;
#pragma unroll
                for (int j = 0; j < nu; j++) v = (ridx == j) ? Bi[i + nx * j] : v;
                bimp[i] = v;
            }
        }
    }

    // own column of V: Phi lanes start at e_ridx (V0[A] = vec(I), :178), others at 0 (:177)
    T c[nx];
#pragma unroll
    for (int i = 0; i < nx; i++) c[i] = (role == R_PHI && i == ridx) ? 1.0 : 0.0;

    // derivs_foh (:235-286) for this lane: returns f (all lanes) and the lane's column derivative
    auto derivs = [&](T t, const T (&xs)[nx], const T (&cs)[nx], T (&fx)[nx],
                      T (&dc)[nx]) {
        // linterp on the 2-point grid (helper.jl:107-118), saturating t
        const T tc = fmax(t0, fmin(t1, t));
        const T cc = (t1 - tc) / (t1 - t0);
        T u[nu];
#pragma unroll
        for (int i = 0; i < nu; i++) u[i] = IMP ? (T)0 : cc * u0[i] + ((T)1 - cc) * u1[i];   // IMPULSE: coasting (:321)
        const T sm = (t1 - t) / (t1 - t0);  // :252
        const T sp = (t - t0) / (t1 - t0);  // :253
        T Am[nx * nx], Bmat[nx * nu], Fc[nx * npFa];
        M::dyn(par, t, k + 1, xs, u, pb, fx, Am, Bmat, Fc);  // :256-259
        // r = f - A x - B u - F p  (:262)
        T rr[nx];
#pragma unroll
        for (int i = 0; i < nx; i++) {
            T acc = fx[i];
#pragma unroll
            for (int j = 0; j < nx; j++) acc -= Am[i + nx * j] * xs[j];
#pragma unroll
            for (int j = 0; j < nu; j++) acc -= Bmat[i + nx * j] * u[j];
            if (npF > 0) {
#pragma unroll
                for (int j = 0; j < npFa; j++) acc -= Fc[i + nx * j] * pF[j];
            }
            rr[i] = acc;
        }
        // working column: Phi lanes carry their Phi column, the others their right-hand side
        T w[nx];
#pragma unroll
        for (int i = 0; i < nx; i++) {
            T v = 0.0;
            if (role == R_PHI) v = cs[i];
            else if (role == R_R) v = rr[i];
            else if (role == R_E) v = (i == ridx) ? 1.0 : 0.0;  // E = I(nx), scp.jl:149
            else if (role == R_BM || role == R_BP) {
                T bcol = 0.0;
#pragma unroll
                for (int j = 0; j < nu; j++) bcol = (ridx == j) ? Bmat[i + nx * j] : bcol;
                v = IMP ? 0.0 : (role == R_BM ? sm : sp) * bcol;  // :260-261 (IMPULSE: no B blocks in V)
            } else if (role == R_F) {
#pragma unroll
                for (int j = 0; j < npFa; j++) v = (ridx == j) ? Fc[i + nx * j] : v;
            }
            w[i] = v;
        }
        // ---- cooperative LU with partial pivoting on [Phi | rhs]  (Phi \ I, :267) ----
        // Structure the model declares (round 5, M::lu_lead / M::lu_decoupled): the first LEAD columns of Phi stay unit upper
        // triangular for the whole interval -- states no dynamics row depends on (Starship: the position, A[:, r] = 0, so
        // Phi[:, r] = e_r exactly; free-flyer: position and velocity, Phi_rv = [I, a I; 0, I]) -- and these entries are EXACT in
        // floating point (products with compile-time zeros are dropped, 0 * x = 0 otherwise).  Elimination step s < LEAD therefore
        // finds pivot 1 in place and multipliers that are exactly 0: skipping it changes no bit of the result, and neither does
        // dropping the divisions by the unit pivots and the updates with zero U entries in the back-substitution.  Free-flyer:
        // 57 of the 78 multiplier broadcasts and 42 of the 78 substitution broadcasts per stage are such no-ops.
        constexpr int LEAD = M::lu_lead;
        constexpr bool DEC = M::lu_decoupled;     // rows < LEAD have zeros in the columns >= LEAD as well (block-diagonal Phi)
#pragma unroll
        for (int s = LEAD; s < nx; s++) {
            int piv = s;
            T mx = fabs(w[s]);
#pragma unroll
            for (int i = s + 1; i < nx; i++) {
                const T ai = fabs(w[i]);
                if (ai > mx) { mx = ai; piv = i; }
            }
            piv = group_bcast<G>(piv, gbase, s);
            // (a whole-wavefront group holds piv in a scalar register: rows are exchanged under a scalar branch, and a step that
            //  needs no exchange -- the usual case, Phi stays close to the identity over an interval -- skips the select chain)
            if (G == 32 || piv != s) {     // (32-lane groups: a wave-wide vote before the exchange was tried and did not pay)
#pragma unroll
                for (int i = s + 1; i < nx; i++) {
                    const bool sw = (piv == i);
                    const T ws = w[s], wi = w[i];
                    w[s] = sw ? wi : ws;
                    w[i] = sw ? ws : wi;
                }
            }
            const T inv = pivot_rcp(w[s]);  // meaningful in lane s (LAPACK getf2 scales by the reciprocal)
#pragma unroll
            for (int i = s + 1; i < nx; i++) {
                const T l = group_bcast<G>(w[i] * inv, gbase, s);
                w[i] = (gl == s) ? w[i] : w[i] - l * w[s];
            }
        }
        // ---- back-substitution, column oriented: y = U^-1 (L^-1 P rhs) ----
        T y[nx];
#pragma unroll
        for (int i = 0; i < nx; i++) y[i] = w[i];
#pragma unroll
        for (int j = nx - 1; j >= 0; j--) {
            if (j >= LEAD) {            // (unit pivots in the leading block)
                const T ujj = group_bcast<G>(w[j], gbase, j);
                y[j] = y[j] * pivot_rcp(ujj);
            }
#pragma unroll
            for (int i = (DEC && j >= LEAD) ? LEAD : 0; i < j; i++) {
                const T uij = group_bcast<G>(w[i], gbase, j);
                y[i] -= uij * y[j];
            }
        }
        // ---- column derivative: Phi lanes A*Phi[:,j] (:268), the others Phi^-1 * rhs (:269-273) ----
#pragma unroll
        for (int i = 0; i < nx; i++) {
            T acc = 0.0;
#pragma unroll
            for (int j = 0; j < nx; j++) acc += Am[i + nx * j] * cs[j];
            dc[i] = (role == R_PHI) ? acc : y[i];
        }
    };

    // ---- RK4 over the sub-grid (rk4_generic, helper.jl:483-498; rk4_core_step :411-424) ----
    for (int j = 1; j < a.Nsub; j++) {
        const T ta = (T)linrange((double)t0, (double)t1, a.Nsub, j - 1);  // LinRange(t[k], t[k+1], Nsub), :197
        const T tb = (T)linrange((double)t0, (double)t1, a.Nsub, j);
        const T h = tb - ta;
        // The four stage evaluations are ONE loop body (round 5): inlined four times the body of this loop was ~140 KB of code for the
        // free-flyer (17.6 k instructions per step) -- twice the 64 KB instruction cache two CUs share -- and one box in round 4 ran
        // the same binary at twice its usual time with 2.7 x the wait cycles.  Same operations in the same order as rk4_core_step
        // (helper.jl:411-424): k1 .. k4 at (ta, ta + h/2, ta + h/2, ta + h), V + h/6 (k1 + 2 k2 + 2 k3 + k4).
        T kx[nx], kc[nx], xs[nx], cs[nx], sx[nx], sc[nx];
#pragma unroll
        for (int i = 0; i < nx; i++) { xs[i] = x[i]; cs[i] = c[i]; sx[i] = 0.0; sc[i] = 0.0; }
#pragma unroll 1
        for (int st = 0; st < 4; st++) {
            const T ts = st == 0 ? ta : (st == 3 ? ta + h : ta + h / 2);
            derivs(ts, xs, cs, kx, kc);
            if (st == 3) {
#pragma unroll
                for (int i = 0; i < nx; i++) {
                    x[i] = x[i] + h / 6 * (sx[i] + kx[i]);
                    c[i] = c[i] + h / 6 * (sc[i] + kc[i]);
                }
            } else {
                const T hs = st == 2 ? h : h / 2;
                if (st == 0) {
#pragma unroll
                    for (int i = 0; i < nx; i++) { sx[i] = kx[i]; sc[i] = kc[i]; }
                } else {
#pragma unroll
                    for (int i = 0; i < nx; i++) { sx[i] += 2 * kx[i]; sc[i] += 2 * kc[i]; }
                }
#pragma unroll
                for (int i = 0; i < nx; i++) { xs[i] = x[i] + hs * kx[i]; cs[i] = c[i] + hs * kc[i]; }
            }
        }
        M::action(x);  // integration actions on the state (helper.jl:494-496)
    }

    // ---- set_update_matrices (:381-403): non-Phi columns are pre-multiplied by Phi(t_{k+1}) ----
    T out[nx];
#pragma unroll
    for (int i = 0; i < nx; i++) out[i] = 0.0;
    T mul[nx];   // the column Phi(t_{k+1}) multiplies: this lane's own, or (IMPULSE, B- lanes) the impulse input matrix
#pragma unroll
    for (int i = 0; i < nx; i++) mul[i] = (IMP && role == R_BM) ? bimp[i] : c[i];
#pragma unroll
    for (int l = 0; l < nx; l++) {
#pragma unroll
        for (int i = 0; i < nx; i++) {
            const T phi_il = group_bcast<G>(c[i], gbase, l);
            out[i] += phi_il * mul[l];
        }
    }
    if (!write) return;
    const long ik = (long)b * (a.N - 1) + k;
    double* dst = nullptr;
    if (role == R_PHI) dst = a.A + (ik * nx + ridx) * nx;
    else if (role == R_BM) dst = a.Bm + (ik * nu + ridx) * nx;
    else if (role == R_BP) dst = a.Bp + (ik * nu + ridx) * nx;
    else if (role == R_F) dst = a.F + (ik * npFa + ridx) * nx;
    else if (role == R_R) dst = a.r + ik * nx;
    else if (role == R_E) dst = a.E + (ik * nx + ridx) * nx;
    if (dst != nullptr) {
#pragma unroll
        for (int i = 0; i < nx; i++) dst[i] = (role == R_PHI) ? c[i] : out[i];
    }
    // defect and feasibility (:205-210)
    if (gl == 0) {
        const double* xn = xk + nx;
        double nrm = 0.0;
#pragma unroll
        for (int i = 0; i < nx; i++) {
            const double d = xn[i] - (double)x[i];
            a.defect[ik * nx + i] = d;
            nrm = fmax(nrm, fabs(a.iSx[i] * d));
        }
        if (nrm > a.feas_tol) atomicAnd(&a.feas[b], 0);
    }
}

// ------------------------------------------------------------------------------------------------
// K1v: variational form of the same quantities, for models whose dynamics are linear in (x, u) with
// coefficient matrices that do not depend on (t, x, u) inside an interval (M::const_jacobian: every registered
// model so far -- linear dynamics scaled by the time dilation).  Instead of integrating int Phi^-1 [..] and
// multiplying by Phi(t_{k+1}) at the end (reference formulation, kernel above), each output column
// Psi = Phi * int Phi^-1 rhs is integrated directly:
//        Psi' = A Psi + rhs(t),  Psi(t_k) = 0      (Phi' = A Phi, Phi(t_k) = I),
// which SURVEY.md App. A explicitly allows when parity holds.  For constant A both forms are the same
// polynomial-in-h RK4 update of the same linear system evaluated at the same stage points, so they agree to
// fp64 round-off (measured <= 5e-16 relative against the oracle for all three models and enforced at 1e-10 in
// tests/test_discretize_gpu.py); for state-dependent Jacobians they would differ by the RK4 truncation
// error, hence the trait (and SCP_DISC_REFERENCE_FORM=1 forces the kernel above).
//
// Mapping: one THREAD per (problem, interval, column); blockIdx.y = column, so a block is role-uniform: no
// divergence, no cross-lane traffic, no LU, no pivot divisions.  Two instantiations:
//   HEAVY = false: columns of Phi, B-, B+, E.  They need neither the state nor the input (A, B constant):
//                  a stage is one structured A*c product (M::Amul) plus an axpy -- ~70 VGPRs, 7 waves/SIMD.
//   HEAVY = true : columns of F and r, which need f(x(t), u(t)): these threads also integrate the state and
//                  write the defect / feasibility flag.
// ------------------------------------------------------------------------------------------------
template <class M, bool HEAVY>
__global__ __launch_bounds__(256) void discretize_foh_var_kernel(DiscArgs a, typename M::Params par)
{
    constexpr int nx = M::nx, nu = M::nu, npF = M::npF;
    constexpr int npFa = npF > 0 ? npF : 1;
    const long total = (long)a.B * (a.N - 1);
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int b = (int)(gid / (a.N - 1));
    const int k = (int)(gid % (a.N - 1));
    if (a.mask != nullptr && a.mask[b] == 0) return;
    const int gl = blockIdx.y;
    const double* pb = a.p + (long)b * np_total<M>(a.N);   // p = [global; node parameters]: the dynamics read the globals
    const double t0 = linrange(0.0, 1.0, a.N, k);
    const double t1 = linrange(0.0, 1.0, a.N, k + 1);
    const long ik = (long)b * (a.N - 1) + k;

    if constexpr (!HEAVY) {
        // ---- light columns: [Phi (nx) | B- (nu) | B+ (nu) | E (nx)] ----
        int role, ridx;
        if (gl < nx) { role = R_PHI; ridx = gl; }
        else if (gl < nx + nu) { role = R_BM; ridx = gl - nx; }
        else if (gl < nx + 2 * nu) { role = R_BP; ridx = gl - nx - nu; }
        else { role = R_E; ridx = gl - nx - 2 * nu; }
        double c[nx], g0[nx];   // g0: constant part of the forcing (column of B, or e_ridx for E)
#pragma unroll
        for (int i = 0; i < nx; i++) { c[i] = (role == R_PHI && i == ridx) ? 1.0 : 0.0; g0[i] = (role == R_E && i == ridx) ? 1.0 : 0.0; }
        if (role == R_BM || role == R_BP) M::Bcol(par, pb, ridx, g0);
        auto derivs = [&](double t, const double (&cs)[nx], double (&dc)[nx]) {
            const double sm = (t1 - t) / (t1 - t0);  // :252
            const double sp = (t - t0) / (t1 - t0);  // :253
            const double sg = role == R_BM ? sm : (role == R_BP ? sp : (role == R_E ? 1.0 : 0.0));
            double ac[nx];
            M::Amul(par, pb, cs, ac);
#pragma unroll
            for (int i = 0; i < nx; i++) dc[i] = ac[i] + sg * g0[i];
        };
        for (int j = 1; j < a.Nsub; j++) {
            const double ta = linrange(t0, t1, a.Nsub, j - 1);
            const double tb = linrange(t0, t1, a.Nsub, j);
            const double h = tb - ta;
            double kc[nx], cs[nx], sc[nx];
            derivs(ta, c, kc);
#pragma unroll
            for (int i = 0; i < nx; i++) { sc[i] = kc[i]; cs[i] = c[i] + h / 2 * kc[i]; }
            derivs(ta + h / 2, cs, kc);
#pragma unroll
            for (int i = 0; i < nx; i++) { sc[i] += 2 * kc[i]; cs[i] = c[i] + h / 2 * kc[i]; }
            derivs(ta + h / 2, cs, kc);
#pragma unroll
            for (int i = 0; i < nx; i++) { sc[i] += 2 * kc[i]; cs[i] = c[i] + h * kc[i]; }
            derivs(ta + h, cs, kc);
#pragma unroll
            for (int i = 0; i < nx; i++) c[i] = c[i] + h / 6 * (sc[i] + kc[i]);
        }
        double* dst;
        if (role == R_PHI) dst = a.A + (ik * nx + ridx) * nx;
        else if (role == R_BM) dst = a.Bm + (ik * nu + ridx) * nx;
        else if (role == R_BP) dst = a.Bp + (ik * nu + ridx) * nx;
        else dst = a.E + (ik * nx + ridx) * nx;
#pragma unroll
        for (int i = 0; i < nx; i++) dst[i] = c[i];
    } else {
        // ---- heavy columns: [F (npF) | r], with the state integrated alongside ----
        const bool isR = gl == npF;
        const int ridx = isR ? 0 : gl;
        const double* xk = a.xd + ((long)b * a.N + k) * nx;
        const double* uk = a.ud + ((long)b * a.N + k) * nu;
        double x[nx], u0[nu], u1[nu], pF[npFa], c[nx];
#pragma unroll
        for (int i = 0; i < nx; i++) { x[i] = xk[i]; c[i] = 0.0; }
#pragma unroll
        for (int i = 0; i < nu; i++) { u0[i] = uk[i]; u1[i] = uk[nu + i]; }
#pragma unroll
        for (int j = 0; j < npFa; j++) pF[j] = (npF > 0) ? pb[M::Fcol(j)] : 0.0;
        auto derivs = [&](double t, const double (&xs)[nx], const double (&cs)[nx], double (&fx)[nx], double (&dc)[nx]) {
            const double tc = fmax(t0, fmin(t1, t));
            const double cc = (t1 - tc) / (t1 - t0);
            double u[nu];
#pragma unroll
            for (int i = 0; i < nu; i++) u[i] = cc * u0[i] + (1.0 - cc) * u1[i];
            double Am[nx * nx], Bmat[nx * nu], Fc[nx * npFa];
            M::dyn(par, t, k + 1, xs, u, pb, fx, Am, Bmat, Fc);
            double rhs[nx], ax[nx], ac[nx];
            M::Amul(par, pb, cs, ac);
            if (isR) {   // r = f - A x - B u - F p  (:262)
                M::Amul(par, pb, xs, ax);
#pragma unroll
                for (int i = 0; i < nx; i++) {
                    double acc = fx[i] - ax[i];
#pragma unroll
                    for (int j = 0; j < nu; j++) acc -= Bmat[i + nx * j] * u[j];
                    if (npF > 0) {
#pragma unroll
                        for (int j = 0; j < npFa; j++) acc -= Fc[i + nx * j] * pF[j];
                    }
                    rhs[i] = acc;
                }
            } else {
#pragma unroll
                for (int i = 0; i < nx; i++) {
                    double v = 0.0;
#pragma unroll
                    for (int j = 0; j < npFa; j++) v = (ridx == j) ? Fc[i + nx * j] : v;
                    rhs[i] = v;
                }
            }
#pragma unroll
            for (int i = 0; i < nx; i++) dc[i] = ac[i] + rhs[i];
        };
        for (int j = 1; j < a.Nsub; j++) {
            const double ta = linrange(t0, t1, a.Nsub, j - 1);
            const double tb = linrange(t0, t1, a.Nsub, j);
            const double h = tb - ta;
            double k1x[nx], k1c[nx], xs[nx], cs[nx], sx[nx], sc[nx];
            derivs(ta, x, c, k1x, k1c);
#pragma unroll
            for (int i = 0; i < nx; i++) { sx[i] = k1x[i]; sc[i] = k1c[i]; xs[i] = x[i] + h / 2 * k1x[i]; cs[i] = c[i] + h / 2 * k1c[i]; }
            derivs(ta + h / 2, xs, cs, k1x, k1c);
#pragma unroll
            for (int i = 0; i < nx; i++) { sx[i] += 2 * k1x[i]; sc[i] += 2 * k1c[i]; xs[i] = x[i] + h / 2 * k1x[i]; cs[i] = c[i] + h / 2 * k1c[i]; }
            derivs(ta + h / 2, xs, cs, k1x, k1c);
#pragma unroll
            for (int i = 0; i < nx; i++) { sx[i] += 2 * k1x[i]; sc[i] += 2 * k1c[i]; xs[i] = x[i] + h * k1x[i]; cs[i] = c[i] + h * k1c[i]; }
            derivs(ta + h, xs, cs, k1x, k1c);
#pragma unroll
            for (int i = 0; i < nx; i++) { x[i] = x[i] + h / 6 * (sx[i] + k1x[i]); c[i] = c[i] + h / 6 * (sc[i] + k1c[i]); }
            M::action(x);  // integration actions on the state (helper.jl:494-496)
        }
        double* dst = isR ? a.r + ik * nx : a.F + (ik * npFa + ridx) * nx;
#pragma unroll
        for (int i = 0; i < nx; i++) dst[i] = c[i];
        if (isR) {   // defect and feasibility (:205-210)
            const double* xn = xk + nx;
            double nrm = 0.0;
#pragma unroll
            for (int i = 0; i < nx; i++) {
                const double d = xn[i] - x[i];
                a.defect[ik * nx + i] = d;
                nrm = fmax(nrm, fabs(a.iSx[i] * d));
            }
            if (nrm > a.feas_tol) atomicAnd(&a.feas[b], 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1x: the variational form for models with STATE-DEPENDENT Jacobians (free-flyer, Starship).  Every output column
// Psi = Phi * int Phi^-1 rhs obeys Psi' = A(t, x(t), u(t)) Psi + rhs(t), so a thread that integrates the state alongside
// needs no Phi^-1: no LU, no cross-lane traffic -- one thread per (problem, interval, column), one instantiation per ROLE
// (the role decides which of f, A, B, F the model evaluation has to produce; the rest is dead code in that instantiation).
//
// Unlike the constant-Jacobian case the two forms are NOT the same polynomial: they differ by RK4 truncation terms
// O((tdil h)^4) (measured on the free-flyer, tests/test_freeflyer_gpu.py: B-, B+ 3.5e-11 at tdil h = 0.05 s, 2.6e-10 at
// 0.072 s; every other block below 1e-12 -- the sigma(t) ramp of the input columns is what the two RK4 schemes see
// differently).  The form is therefore chosen PER PROBLEM: disc_split_kernel sends a problem here when its physical step
// M::time_dilation(p) * h is at most M::var_form_max_phys_step (1e-10 parity with the reference formulation), to the
// reference-form kernel K1 otherwise.
// ------------------------------------------------------------------------------------------------
template <class M, int ROLE>
__global__ __launch_bounds__(256) void discretize_foh_varx_kernel(DiscArgs a, typename M::Params par)
{
    constexpr int nx = M::nx, nu = M::nu, npF = M::npF, npFa = npF > 0 ? npF : 1;
    const long total = (long)a.B * (a.N - 1);
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int b = (int)(gid / (a.N - 1));
    const int k = (int)(gid % (a.N - 1));
    if (a.mask != nullptr && a.mask[b] == 0) return;
    const int ridx = blockIdx.y;                      // column inside the role's block
    const double* pb = a.p + (long)b * np_total<M>(a.N);
    const double t0 = linrange(0.0, 1.0, a.N, k);
    const double t1 = linrange(0.0, 1.0, a.N, k + 1);
    const long ik = (long)b * (a.N - 1) + k;
    const double* xk = a.xd + ((long)b * a.N + k) * nx;
    const double* uk = a.ud + ((long)b * a.N + k) * nu;
    double x[nx], c[nx], u0[nu], u1[nu], pF[npFa];
#pragma unroll
    for (int i = 0; i < nx; i++) { x[i] = xk[i]; c[i] = (ROLE == R_PHI && i == ridx) ? 1.0 : 0.0; }
#pragma unroll
    for (int i = 0; i < nu; i++) { u0[i] = uk[i]; u1[i] = uk[nu + i]; }
#pragma unroll
    for (int j = 0; j < npFa; j++) pF[j] = (npF > 0) ? pb[M::Fcol(j)] : 0.0;
    auto derivs = [&](double t, const double (&xs)[nx], const double (&cs)[nx], double (&fx)[nx], double (&dc)[nx]) {
        const double tc = fmax(t0, fmin(t1, t));
        const double cc = (t1 - tc) / (t1 - t0);
        double u[nu];
#pragma unroll
        for (int i = 0; i < nu; i++) u[i] = cc * u0[i] + (1.0 - cc) * u1[i];
        double Am[nx * nx], Bmat[nx * nu], Fc[nx * npFa];
        M::dyn(par, t, k + 1, xs, u, pb, fx, Am, Bmat, Fc);       // :256-259 (unused outputs are dead code per ROLE)
        double rhs[nx];
        if constexpr (ROLE == R_BM || ROLE == R_BP) {
            const double sg = ROLE == R_BM ? (t1 - t) / (t1 - t0) : (t - t0) / (t1 - t0);   // :252-253
#pragma unroll
            for (int i = 0; i < nx; i++) {
                double v = 0.0;
#pragma unroll
                for (int j = 0; j < nu; j++) v = (ridx == j) ? Bmat[i + nx * j] : v;
                rhs[i] = sg * v;
            }
        } else if constexpr (ROLE == R_F) {
#pragma unroll
            for (int i = 0; i < nx; i++) {
                double v = 0.0;
#pragma unroll
                for (int j = 0; j < npFa; j++) v = (ridx == j) ? Fc[i + nx * j] : v;
                rhs[i] = v;
            }
        } else if constexpr (ROLE == R_R) {       // r = f - A x - B u - F p  (:262)
            double ax[nx];
            if constexpr (M::has_amulx) M::Amulx(par, pb, xs, xs, ax);
            else {
#pragma unroll
                for (int i = 0; i < nx; i++) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < nx; j++) acc += Am[i + nx * j] * xs[j];
                    ax[i] = acc;
                }
            }
#pragma unroll
            for (int i = 0; i < nx; i++) {
                double acc = fx[i] - ax[i];
#pragma unroll
                for (int j = 0; j < nu; j++) acc -= Bmat[i + nx * j] * u[j];
                if (npF > 0) {
#pragma unroll
                    for (int j = 0; j < npFa; j++) acc -= Fc[i + nx * j] * pF[j];
                }
                rhs[i] = acc;
            }
        } else {
#pragma unroll
            for (int i = 0; i < nx; i++) rhs[i] = (ROLE == R_E && i == ridx) ? 1.0 : 0.0;
        }
        if constexpr (M::has_amulx) {
            double ac[nx];
            M::Amulx(par, pb, xs, cs, ac);
#pragma unroll
            for (int i = 0; i < nx; i++) dc[i] = rhs[i] + ac[i];
        } else {
#pragma unroll
            for (int i = 0; i < nx; i++) {
                double acc = rhs[i];
#pragma unroll
                for (int j = 0; j < nx; j++) acc += Am[i + nx * j] * cs[j];
                dc[i] = acc;
            }
        }
    };
    for (int j = 1; j < a.Nsub; j++) {
        const double ta = linrange(t0, t1, a.Nsub, j - 1);
        const double tb = linrange(t0, t1, a.Nsub, j);
        const double h = tb - ta;
        double k1x[nx], k1c[nx], xs[nx], cs[nx], sx[nx], sc[nx];
        derivs(ta, x, c, k1x, k1c);
#pragma unroll
        for (int i = 0; i < nx; i++) { sx[i] = k1x[i]; sc[i] = k1c[i]; xs[i] = x[i] + h / 2 * k1x[i]; cs[i] = c[i] + h / 2 * k1c[i]; }
        derivs(ta + h / 2, xs, cs, k1x, k1c);
#pragma unroll
        for (int i = 0; i < nx; i++) { sx[i] += 2 * k1x[i]; sc[i] += 2 * k1c[i]; xs[i] = x[i] + h / 2 * k1x[i]; cs[i] = c[i] + h / 2 * k1c[i]; }
        derivs(ta + h / 2, xs, cs, k1x, k1c);
#pragma unroll
        for (int i = 0; i < nx; i++) { sx[i] += 2 * k1x[i]; sc[i] += 2 * k1c[i]; xs[i] = x[i] + h * k1x[i]; cs[i] = c[i] + h * k1c[i]; }
        derivs(ta + h, xs, cs, k1x, k1c);
#pragma unroll
        for (int i = 0; i < nx; i++) { x[i] = x[i] + h / 6 * (sx[i] + k1x[i]); c[i] = c[i] + h / 6 * (sc[i] + k1c[i]); }
        M::action(x);  // integration actions on the state (helper.jl:494-496)
    }
    double* dst;
    if (ROLE == R_PHI) dst = a.A + (ik * nx + ridx) * nx;
    else if (ROLE == R_BM) dst = a.Bm + (ik * nu + ridx) * nx;
    else if (ROLE == R_BP) dst = a.Bp + (ik * nu + ridx) * nx;
    else if (ROLE == R_F) dst = a.F + (ik * npFa + ridx) * nx;
    else if (ROLE == R_R) dst = a.r + ik * nx;
    else dst = a.E + (ik * nx + ridx) * nx;
#pragma unroll
    for (int i = 0; i < nx; i++) dst[i] = c[i];
    if (ROLE == R_R) {   // defect and feasibility (:205-210)
        const double* xn = xk + nx;
        double nrm = 0.0;
#pragma unroll
        for (int i = 0; i < nx; i++) {
            const double d = xn[i] - x[i];
            a.defect[ik * nx + i] = d;
            nrm = fmax(nrm, fabs(a.iSx[i] * d));
        }
        if (nrm > a.feas_tol) atomicAnd(&a.feas[b], 0);
    }
}

// per-problem choice of the form: mvar[b] = 1 -> K1x, mref[b] = 1 -> K1 (both 0 for problems masked out by the SCP loop)
template <class M>
__global__ void disc_split_kernel(int B, int N, int Nsub, const double* p, const int* mask, typename M::Params par, int force_ref,
                                  int* mvar, int* mref)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const bool live = mask == nullptr || mask[b] != 0;
    const double h = 1.0 / ((double)(N - 1) * (double)(Nsub - 1));
    const bool var = !force_ref && M::time_dilation(par, p + (long)b * np_total<M>(N)) * h <= M::var_form_max_phys_step;
    mvar[b] = live && var ? 1 : 0;
    mref[b] = live && !var ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// propagate (FOH), src/solvers/discretization.jl:515-541: integrate the NONLINEAR dynamics from xd[:,1] over
// tc = LinRange(0,1,res) with u(t) = linterp(ud, t_grid) and one classic RK4 step between consecutive tc
// (rk4(...; full=true), helper.jl:483-498).  Serial in time, independent across the batch: one thread per problem.
// The reference's node index k(t) = max(floor(t/(N-1))+1, N) is N for every t.
// ------------------------------------------------------------------------------------------------
struct PropArgs {
    int B, N, res;
    const double* xd;  // [nx,N,B]   (only the first node is read)
    const double* ud;  // [nu,N,B]
    const double* p;   // [np,B]
    double* xc;        // [nx,res,B]
};

template <class M>
__global__ __launch_bounds__(64) void propagate_foh_kernel(PropArgs a, typename M::Params par)
{
    constexpr int nx = M::nx, nu = M::nu, npF = M::npF;
    constexpr int npFa = npF > 0 ? npF : 1;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const double* ub = a.ud + (long)b * a.N * nu;
    const double* pb = a.p + (long)b * np_total<M>(a.N);   // p = [global; node parameters]: the dynamics read the globals
    double* xo = a.xc + (long)b * a.res * nx;
    double x[nx];
#pragma unroll
    for (int i = 0; i < nx; i++) { x[i] = a.xd[(long)b * a.N * nx + i]; xo[i] = x[i]; }
    // u(t): linterp on t_grid = LinRange(0,1,N) (helper.jl:107-118) with get_interval (:84-90): bin = number of grid
    // points strictly below t (at least 1); the candidate from floor() is corrected against the exact grid values
    auto input = [&](double t, double (&u)[nu]) {
        const double g0 = linrange(0.0, 1.0, a.N, 0), g1 = linrange(0.0, 1.0, a.N, a.N - 1);
        t = fmax(g0, fmin(g1, t));
        int k = (int)floor(t * (a.N - 1));
        k = k < 0 ? 0 : (k > a.N ? a.N : k);
        while (k < a.N && t > linrange(0.0, 1.0, a.N, k)) k++;
        while (k > 0 && !(t > linrange(0.0, 1.0, a.N, k - 1))) k--;
        if (k == 0) k = 1;
        const double ta = linrange(0.0, 1.0, a.N, k - 1), tb = linrange(0.0, 1.0, a.N, k);
        const double c = (tb - t) / (tb - ta);
#pragma unroll
        for (int i = 0; i < nu; i++) u[i] = c * ub[(long)(k - 1) * nu + i] + (1.0 - c) * ub[(long)k * nu + i];
    };
    auto f = [&](double t, const double (&xs)[nx], double (&fx)[nx]) {
        double u[nu], Am[nx * nx], Bmat[nx * nu], Fc[nx * npFa];
        input(t, u);
        M::dyn(par, t, a.N, xs, u, pb, fx, Am, Bmat, Fc);   // only f survives dead-code elimination
    };
    for (int j = 1; j < a.res; j++) {
        const double t = linrange(0.0, 1.0, a.res, j - 1), tp = linrange(0.0, 1.0, a.res, j), h = tp - t;
        double k1[nx], k2[nx], k3[nx], k4[nx], tmp[nx];
        f(t, x, k1);
#pragma unroll
        for (int i = 0; i < nx; i++) tmp[i] = x[i] + h / 2 * k1[i];
        f(t + h / 2, tmp, k2);
#pragma unroll
        for (int i = 0; i < nx; i++) tmp[i] = x[i] + h / 2 * k2[i];
        f(t + h / 2, tmp, k3);
#pragma unroll
        for (int i = 0; i < nx; i++) tmp[i] = x[i] + h * k3[i];
        f(t + h, tmp, k4);
#pragma unroll
        for (int i = 0; i < nx; i++) x[i] = x[i] + h / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        M::action(x);
#pragma unroll
        for (int i = 0; i < nx; i++) xo[(long)j * nx + i] = x[i];
    }
}

// ------------------------------------------------------------------------------------------------
// propagate (IMPULSE), src/solvers/discretization.jl:542-560: every interval restarts from its node, x0 = xd[:,k] +
// f(t_k, -k, xd[:,k], ud[:,k], p) (the model's impulse response), and coasts with idle inputs over
// LinRange(t_k, t_{k+1}, subres), subres = ceil(res / (N - 1)).  The intervals are independent: one thread per
// (problem, interval).  xc[nx, 1 + (N-1) subres, B]: sample 0 = xd[:,1], then the subres samples of every interval.
// ------------------------------------------------------------------------------------------------
struct PropImpArgs {
    int B, N, sub;
    const double* xd;  // [nx,N,B]
    const double* ud;  // [nu,N,B]
    const double* p;   // [np,B]
    double* xc;        // [nx, 1 + (N-1) sub, B]
};

template <class M>
__global__ __launch_bounds__(64) void propagate_impulse_kernel(PropImpArgs a, typename M::Params par)
{
    constexpr int nx = M::nx, nu = M::nu, npF = M::npF;
    constexpr int npFa = npF > 0 ? npF : 1;
    const long gid = (long)blockIdx.x * 64 + threadIdx.x;
    if (gid >= (long)a.B * (a.N - 1)) return;
    const int b = (int)(gid / (a.N - 1)), k = (int)(gid % (a.N - 1));
    const double* pb = a.p + (long)b * np_total<M>(a.N);
    const long ns = 1 + (long)(a.N - 1) * a.sub;
    double* xo = a.xc + ((long)b * ns + 1 + (long)k * a.sub) * nx;
    const double t0 = linrange(0.0, 1.0, a.N, k), t1 = linrange(0.0, 1.0, a.N, k + 1);
    double x[nx], u[nu], dx[nx], Bi[nx * nu];
#pragma unroll
    for (int i = 0; i < nx; i++) x[i] = a.xd[((long)b * a.N + k) * nx + i];
#pragma unroll
    for (int i = 0; i < nu; i++) u[i] = a.ud[((long)b * a.N + k) * nu + i];
    if (k == 0) {
#pragma unroll
        for (int i = 0; i < nx; i++) a.xc[(long)b * ns * nx + i] = x[i];          // xc_intvl[1] = xd[:, 1]
    }
    M::impulse(par, t0, k + 1, x, u, pb, dx, Bi);                                  // f(td[k], -k, ...)  (:549)
#pragma unroll
    for (int i = 0; i < nx; i++) { x[i] += dx[i]; xo[i] = x[i]; }
#pragma unroll
    for (int i = 0; i < nu; i++) u[i] = 0.0;                                       // u_idle (:550)
    auto f = [&](double t, const double (&xs)[nx], double (&fx)[nx]) {
        double Am[nx * nx], Bmat[nx * nu], Fc[nx * npFa];
        M::dyn(par, t, a.N, xs, u, pb, fx, Am, Bmat, Fc);
    };
    for (int j = 1; j < a.sub; j++) {
        const double t = linrange(t0, t1, a.sub, j - 1), tp = linrange(t0, t1, a.sub, j), h = tp - t;
        double k1[nx], k2[nx], k3[nx], k4[nx], tmp[nx];
        f(t, x, k1);
#pragma unroll
        for (int i = 0; i < nx; i++) tmp[i] = x[i] + h / 2 * k1[i];
        f(t + h / 2, tmp, k2);
#pragma unroll
        for (int i = 0; i < nx; i++) tmp[i] = x[i] + h / 2 * k2[i];
        f(t + h / 2, tmp, k3);
#pragma unroll
        for (int i = 0; i < nx; i++) tmp[i] = x[i] + h * k3[i];
        f(t + h, tmp, k4);
#pragma unroll
        for (int i = 0; i < nx; i++) x[i] = x[i] + h / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        M::action(x);
#pragma unroll
        for (int i = 0; i < nx; i++) xo[(long)j * nx + i] = x[i];
    }
}

}  // namespace scp
