// Generic batched conic solver, host side: SYMBOLIC analysis of the shared sparsity pattern.
//
// The reference's only true plugin seam is the convex solver behind `ConicProgram` (src/parser/program.jl:63-76,
// `solve!` -> JuMP.optimize! :419-424), which reaches libecos with the standard form
//
//     min 1/2 x'Px + c'x   s.t.  A x = b,   G x + s = h,   s in K = R+^l x Q^{q_1} x ... x Q^{q_nc}
//
// (ECOS form + native quadratic cost instead of MOI's quadratic->SOC bridge).  In an SCP run every subproblem of a
// Monte-Carlo batch -- and every iteration -- has the SAME sparsity pattern; only the values change.  What ECOS does
// once per solve (AMD ordering + symbolic LDL' of the KKT matrix, `ECOS_setup`) is done here once per pattern, on
// the host, and turned into a static SCHEDULE the device kernel (conic_ipm.hpp) replays for every problem:
//
//     KKT (permuted by a minimum-degree ordering)     [ P + dI    A'     Gt'     ]      Gt = W^-1 G
//                                                     [ A        -dI             ]
//                                                     [ Gt               -(1+d)I ]
//
//   * the pattern of Gt: rows of one second-order cone are unioned per column (W^-1 is dense inside a cone);
//   * L (unit lower, CSC) and, for every entry L(i,j), the list of PAIRS (pos L(i,k), pos L(j,k)), k < j, whose
//     products it accumulates -- so the numeric phase is a stream of fused multiply-adds on register accumulators
//     with no index arithmetic, no scatter and no workspace;
//   * row lists (CSR view of L) for the forward substitution, CSR views of A, Gt, P for the mat-vecs.
//
// Everything here is plain host C++ (no HIP).
#pragma once
#include <future>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <numeric>
#include <set>
#include <stdexcept>
#include <vector>

namespace scp {
namespace conic {

struct Csc {
    int nrow = 0, ncol = 0;
    std::vector<int> p, i;   // column pointers [ncol+1], row indices [nnz] (sorted within a column, no duplicates)
    int nnz() const { return p.empty() ? 0 : p.back(); }
};

// row view of a CSC matrix: for every row the (column, position-in-CSC) list
struct CsrView {
    std::vector<int> p, j, pos;
};
inline CsrView csr_view(const Csc& M)
{
    CsrView R;
    R.p.assign(M.nrow + 1, 0);
    for (int e = 0; e < M.nnz(); e++) R.p[M.i[e] + 1]++;
    for (int r = 0; r < M.nrow; r++) R.p[r + 1] += R.p[r];
    R.j.resize(M.nnz()); R.pos.resize(M.nnz());
    std::vector<int> w(R.p.begin(), R.p.end() - 1);
    for (int c = 0; c < M.ncol; c++)
        for (int e = M.p[c]; e < M.p[c + 1]; e++) { const int q = w[M.i[e]]++; R.j[q] = c; R.pos[q] = e; }
    return R;
}

inline void check_csc(const Csc& M, const char* name, bool upper)
{
    if ((int)M.p.size() != M.ncol + 1 || M.p[0] != 0) throw std::invalid_argument(std::string(name) + ": bad column pointers");
    // before any M.i[e] is read: a caller that passes a NULL index array with nnz > 0 leaves M.i empty (make_csc)
    if (M.p[M.ncol] < 0 || (long)M.i.size() != (long)M.p[M.ncol]) throw std::invalid_argument(std::string(name) + ": index array length");
    for (int c = 0; c < M.ncol; c++) {
        if (M.p[c + 1] < M.p[c] || M.p[c + 1] > M.p[M.ncol]) throw std::invalid_argument(std::string(name) + ": column pointers not monotone");
        for (int e = M.p[c]; e < M.p[c + 1]; e++) {
            if (M.i[e] < 0 || M.i[e] >= M.nrow) throw std::invalid_argument(std::string(name) + ": row index out of range");
            if (e > M.p[c] && M.i[e] <= M.i[e - 1]) throw std::invalid_argument(std::string(name) + ": row indices not strictly increasing");
            if (upper && M.i[e] > c) throw std::invalid_argument(std::string(name) + ": entry below the diagonal (upper triangle expected)");
        }
    }
}

// source of a KKT entry: which value array it is copied from
enum Src : int { SRC_NONE = 0, SRC_P = 1, SRC_A = 2, SRC_GT = 3 };

struct Symbolic {
    int n = 0, p = 0, m = 0, l = 0, nk = 0;
    std::vector<int> q, cone_off;      // SOC dimensions, first row of each cone
    std::vector<int> row_cone;         // [m] cone index of a row (-1: R+ row)
    Csc P, A, G, Gt;                   // P: upper triangle; Gt: unioned pattern
    std::vector<int> g2gt;             // [nnzG] position of G entry inside Gt
    // Gt build jobs: for every (column, cone) block of Gt: [gt_first, d) destination range, source entries of G
    std::vector<int> job_gt0, job_cone, job_src_p;   // per job: first Gt position, cone id, pointer into job_src_*
    std::vector<int> job_src_row, job_src_g;         // per source: row inside the cone, position in G
    std::vector<int> lp_gt, lp_g;                    // R+ entries: Gt position <- G position (row = Gt.i)
    CsrView Ar, Gtr, Pfull;                          // row views (Pfull: symmetric expansion; pos into P)
    // rows of the UNREGULARISED scaled KKT matrix [P A' Gt'; A 0 0; Gt 0 -I] as one list per row (round 5): term t of row r is
    // value src_val(kk_src[t], kk_idx[t]) (1: osc Px, 2: Ax, 3: Gt) times sol[kk_col[t]], in the order P, A', Gt' (x rows) -- what
    // kkt_residual streams with four terms in flight instead of three nests of dependent loads
    std::vector<int> kk_p, kk_src, kk_idx, kk_col;
    std::vector<int> kk_long;          // rows with more than KK_LONG terms (the globally coupled variables): summed by a whole wave each
    static constexpr int KK_LONG = 64;
    std::vector<int> Pfull_diag;                     // not used by the kernel; kept for tests
    // ordering and factor
    std::vector<int> perm, iperm;      // perm[new] = old, iperm[old] = new
    std::vector<int> Lp, Li;           // CSC of strict lower L in permuted numbering
    std::vector<int> l_src, l_src_idx; // [nnzL] source of K(i,j) for that entry
    std::vector<int> d_src, d_src_idx; // [nk] source of the diagonal K(j,j) (P diagonal or none)
    std::vector<int> d_kind;           // [nk] 0: +d (x block), 1: -d (y block), 2: -(1+d) (z block)
    std::vector<int64_t> pair_p;       // [nnzL+1]
    std::vector<int> pair_a, pair_b;   // positions of L(i,k) / L(j,k)
    std::vector<int> row_p, row_k, row_pos;   // [nk+1], row lists of L (column k, position)
    int64_t flops = 0;                 // multiply-adds of one numeric factorisation
    int nd_depth = 0;                  // > 0: the ordering is the nested dissection of nd_ranks with this many levels
    int n_free = 0;                    // variables with no cone row and no quadratic cost: their pivots rest on the static
                                       // regularisation alone (see auto_reg)
    // level sets (columns of one level are mutually independent): the device kernel spreads the columns / entries /
    // rows of a level over the waves of a workgroup and synchronises between levels
    std::vector<int> lev_p, lev_cols;          // factor + forward substitution: level(j) = 1 + max level(k), L(j,k) != 0
    std::vector<int> lev_ent_p, lev_ent, ent_col;   // entries of L grouped by the level of their column; column of an entry
    std::vector<int> rlev_p, rlev_cols;        // backward substitution: rlevel(j) = 1 + max rlevel(i), L(i,j) != 0
    // Long items.  A worker owns a whole row (pivot sum, forward substitution) or a whole pair list (entry of L); the rows of
    // the globally coupled variables (time dilation, trust-region epigraphs: ordered last) and the pair lists between them
    // run over nearly every earlier column -- 8 800 terms on the Starship N = 100 program, where the serial chains of one
    // forward sweep added up to 98 500 multiply-adds against 390 for an even split over 1 024 workers (tools/order_survey.py).
    // Items longer than LONG_ITEM are therefore cut into chunks of ~sqrt(length) terms: the chunks of a level are summed by
    // different workers into a scratch vector, a second phase (one more barrier, only on levels that have such items)
    // combines them in a fixed order.  Inside a level the short items come first (lev_nshort / lev_ent_nshort).
    static constexpr int LONG_ITEM = 128;
    std::vector<int> lev_nshort, rchunk_p, rchunk_r0, rchunk_r1, col_c0, col_c1;   // col_c*: by position in lev_cols
    std::vector<int> lev_ent_nshort, echunk_p, ent_c0, ent_c1;                     // ent_c*: by position in lev_ent
    std::vector<int64_t> echunk_q0, echunk_q1;
    int max_chunks = 1;                        // scratch slots a level needs
};

// ---------- minimum-degree ordering on the pattern of a symmetric matrix (adjacency as sorted vectors) ----------
// `cls` (optional): a vertex of a lower class is eliminated before any vertex of a higher class (constrained minimum
// degree); inside a class the usual rule applies.  `unlock` (optional, with cls): class-2 vertices are BLOCKED until
// one of their class-1 neighbours has been eliminated, then they join class 1.
// `rank` (optional, with cls): inside a class, vertices of a lower rank go first (nested dissection: leaves before their
// separators, nd_ranks below).
inline std::vector<int> min_degree(int n, const std::vector<std::vector<int>>& adj0, std::vector<int>* cls = nullptr,
                                   bool unlock = false, const std::vector<int>* rank = nullptr)
{
    // Elimination graph with EXACT degrees (explicit fill), adjacency as sorted vectors: eliminating v replaces the list of every
    // neighbour w by (adj[w] U adj[v]) \ {v, w} -- one linear merge per neighbour.  (Until round 5 the lists were std::set and the
    // clique of adj[v] was inserted pair by pair: the same order, element for element -- the key, the tie-break and the update
    // sequence are unchanged -- but 5 s instead of 0.3 s on the dissected free-flyer N = 200 pattern, whose separators are
    // eliminated with hundreds of neighbours; `create` of that template was 12.7 s of host time, round-5 bench line.)
    std::vector<std::vector<int>> adj(n);
    for (int v = 0; v < n; v++) for (int w : adj0[v]) if (w != v) { adj[v].push_back(w); adj[w].push_back(v); }
    for (int v = 0; v < n; v++) { std::sort(adj[v].begin(), adj[v].end()); adj[v].erase(std::unique(adj[v].begin(), adj[v].end()), adj[v].end()); }
    std::set<std::pair<long long, int>> heap;   // (class * 2^52 + rank * 2^32 + degree, vertex)
    std::vector<long long> deg(n);
    auto key = [&](int v) {
        return (cls ? (long long)(*cls)[v] << 52 : 0LL) + (rank ? (long long)(*rank)[v] << 32 : 0LL) + (long long)adj[v].size();
    };
    for (int v = 0; v < n; v++) { deg[v] = key(v); heap.insert({deg[v], v}); }
    std::vector<int> order; order.reserve(n);
    std::vector<int> nb, merged;
    while (!heap.empty()) {
        const int v = heap.begin()->second;
        heap.erase(heap.begin());
        order.push_back(v);
        nb.swap(adj[v]);
        adj[v].clear();
        for (int w : nb) heap.erase({deg[w], w});
        if (unlock && (*cls)[v] == 1) for (int w : nb) if ((*cls)[w] == 2) (*cls)[w] = 1;
        for (int w : nb) {
            const std::vector<int>& aw = adj[w];
            merged.clear(); merged.reserve(aw.size() + nb.size());
            size_t i = 0, j = 0;
            while (i < aw.size() || j < nb.size()) {
                int x;
                if (j == nb.size() || (i < aw.size() && aw[i] < nb[j])) x = aw[i++];
                else if (i == aw.size() || nb[j] < aw[i]) x = nb[j++];
                else { x = aw[i]; i++; j++; }
                if (x != v && x != w) merged.push_back(x);
            }
            adj[w].swap(merged);
        }
        for (int w : nb) { deg[w] = key(w); heap.insert({deg[w], w}); }
    }
    return order;
}

// ---------- nested-dissection ranks for time-staged programs ----------
// The KKT graph of an SCP subproblem is a CHAIN of node blocks: after the cone rows z are eliminated, the variables of
// one node form a connected block (coupled through P + Gt'Gt), consecutive blocks are joined ONLY by equality rows (the
// dynamics x_{k+1} = A x_k + ...), and a few global variables (time dilation, trust-region epigraphs) touch everything.
// The minimum-degree ordering of such a chain is sequential: the elimination tree is a path, one short level per node
// (713 levels for the rocket at N = 100), and the level-scheduled kernel pays two workgroup barriers per level.
// Here: (1) drop the globally coupled vertices (degree > 4x median; they are ordered last), (2) contract the x-x
// components, (3) breadth-first levels of the component graph whose edges are the equality rows, (4) recursive bisection:
// the separator of a cut is the set of EQUALITY ROWS joining the two sides -- never a variable, so that every variable is
// still eliminated inside its own block with its equality rows pending (the pivots keep the sign-definite structure the
// sequential order has; a variable eliminated after all of its rows would have a cancellation-prone pivot), and the
// separators are eliminated on the negative-definite Schur complement in cyclic-reduction order.  rank = 0 for block
// vertices, larger for separators nearer the root; the caller feeds it to the constrained minimum degree.  Depth of the
// elimination tree: O(block depth + nx log N) instead of O(N block depth) -- 68 levels for the rocket at N = 100 with
// 18 % more multiply-adds.  Returns the number of dissection levels (0: no chain found, ranks all zero).
inline int nd_ranks(int n, int p, const std::vector<std::vector<int>>& adj /* KKT adjacency, sorted */, std::vector<int>& rank,
                    int leaf_levels = 1, double dense_factor = 4.0)
{
    const int nv = n + p, nk = (int)adj.size();
    rank.assign(nk, 0);
    if (p == 0 || n == 0) return 0;
    // graph on x, y after eliminating every z (a z vertex makes its variables a clique)
    std::vector<std::vector<int>> g(nv);
    for (int v = 0; v < nv; v++) for (int w : adj[v]) if (w < nv) g[v].push_back(w);
    for (int z = nv; z < nk; z++) {
        const std::vector<int>& nb = adj[z];
        for (int a : nb) for (int b : nb) if (a != b && a < nv && b < nv) g[a].push_back(b);
    }
    std::vector<int> deg(nv);
    for (int v = 0; v < nv; v++) {
        std::sort(g[v].begin(), g[v].end()); g[v].erase(std::unique(g[v].begin(), g[v].end()), g[v].end());
        deg[v] = (int)g[v].size();
    }
    std::vector<int> sorted_deg(deg);
    std::nth_element(sorted_deg.begin(), sorted_deg.begin() + nv / 2, sorted_deg.end());
    const double thr = std::max(40.0, dense_factor * (double)sorted_deg[nv / 2]);
    std::vector<char> keep(nv);
    for (int v = 0; v < nv; v++) keep[v] = deg[v] <= thr;
    auto is_y = [&](int v) { return v >= n; };
    // components of the x-x graph
    std::vector<int> comp(nv, -1);
    int nc = 0;
    std::vector<int> stack;
    for (int s0 = 0; s0 < n; s0++) {
        if (!keep[s0] || comp[s0] >= 0) continue;
        comp[s0] = nc; stack.assign(1, s0);
        while (!stack.empty()) {
            const int v = stack.back(); stack.pop_back();
            for (int w : g[v]) if (!is_y(w) && keep[w] && comp[w] < 0) { comp[w] = nc; stack.push_back(w); }
        }
        nc++;
    }
    // equality rows as hyper-edges between components
    std::vector<std::vector<int>> ycomps(p), cadj(nc);
    for (int y = 0; y < p; y++) {
        if (!keep[n + y]) continue;
        std::vector<int>& cs = ycomps[y];
        for (int w : g[n + y]) if (!is_y(w) && keep[w]) cs.push_back(comp[w]);
        std::sort(cs.begin(), cs.end()); cs.erase(std::unique(cs.begin(), cs.end()), cs.end());
        for (int c : cs) cadj[c].push_back(y);
    }
    auto bfs = [&](int c0, std::vector<int>& lev) {   // levels of the component graph from c0; returns the last component reached
        std::vector<int> fr(1, c0), nx;
        lev[c0] = 0;
        int last = c0;
        while (!fr.empty()) {
            nx.clear();
            for (int c : fr)
                for (int y : cadj[c])
                    for (int c2 : ycomps[y]) if (lev[c2] < 0) { lev[c2] = lev[c] + 1; nx.push_back(c2); last = c2; }
            fr.swap(nx);
        }
        return last;
    };
    std::vector<char> done(nc, 0);
    struct Sep { std::vector<int> ys; int depth; };
    std::vector<Sep> seps;
    int maxdepth = 0;
    for (int c0 = 0; c0 < nc; c0++) {
        if (done[c0]) continue;
        std::vector<int> lev(nc, -1);
        int far = bfs(c0, lev);
        std::vector<int> members;
        for (int c = 0; c < nc; c++) if (lev[c] >= 0) { members.push_back(c); done[c] = 1; }
        for (int sweep = 0; sweep < 2; sweep++) {   // pseudo-peripheral start
            for (int c : members) lev[c] = -1;
            far = bfs(far, lev);
        }
        int L = 0;
        for (int c : members) L = std::max(L, lev[c] + 1);
        if (L <= leaf_levels) continue;
        std::vector<std::vector<int>> cuts(L);     // cuts[l]: equality rows joining level l and l + 1
        for (int c : members)
            for (int y : cadj[c]) {
                int lo = 1 << 30, hi = -1;
                for (int c2 : ycomps[y]) { lo = std::min(lo, lev[c2]); hi = std::max(hi, lev[c2]); }
                if (hi > lo && lev[c] == lo) cuts[lo].push_back(y);
            }
        struct Job { int lo, hi, depth; };
        std::vector<Job> jobs(1, Job{0, L - 1, 0});
        while (!jobs.empty()) {
            const Job j = jobs.back(); jobs.pop_back();
            if (j.hi - j.lo + 1 <= leaf_levels) continue;
            const int mid = (j.lo + j.hi - 1) / 2;
            std::vector<int>& c = cuts[mid];
            std::sort(c.begin(), c.end()); c.erase(std::unique(c.begin(), c.end()), c.end());
            seps.push_back(Sep{c, j.depth});
            maxdepth = std::max(maxdepth, j.depth + 1);
            jobs.push_back(Job{j.lo, mid, j.depth + 1}); jobs.push_back(Job{mid + 1, j.hi, j.depth + 1});
        }
    }
    if (maxdepth == 0) return 0;
    for (const Sep& s : seps) for (int y : s.ys) rank[n + y] = std::max(rank[n + y], maxdepth - s.depth);
    for (int v = 0; v < nv; v++) if (!keep[v]) rank[v] = maxdepth + 1;
    return maxdepth;
}

enum Ordering : int { ORDER_SEQUENTIAL = 0, ORDER_NESTED = 1 };

// ordering: user_perm, or minimum degree -- by default CONSTRAINED so that no pivot is ever just the static
// regularisation "+-d plus rounding noise" (which an unconstrained ordering produces when it eliminates an equality row
// before any of its variables, or a cost-free variable before any of its rows: pivot +-d, fill of size 1/d, wrong-signed
// pivots a few columns later):
//   1. the cone rows z first (pivots -(1+d): always well conditioned; this accumulates P + Gt'Gt on the variables),
//   2. then variables x and equality multipliers y together by minimum degree, a y being eligible only once one of its
//      variables has been eliminated (its pivot is then -d - a^2/D_x).
// The fill stays within a few percent of the unconstrained ordering (eliminating ALL x before the y would make the
// Schur complement on y dense).  free_order = true gives the plain rule.
// Static regularisation when the caller leaves it to the solver (opts.reg < 0).  Programs in which every variable sits in
// a cone row or has a quadratic cost (all SCP subproblems: trust-region rows) get 1e-10: the pivots are dominated by
// P + Gt'Gt.  (1e-8 until round 5.  On the equality block the refinement against the unregularised matrix contracts by
// reg / (A H^-1 A'), and late in a run H = P + Gt'Gt carries 1e12 on the active rows: with 1e-8 the GuSTO programs needed 6 - 8
// refinement steps per IPM iteration and the primal residual stalled at 4e-9 -- inside ECOS's feastol, but worth 6e-6 of the
// optimal value once multiplied by the multipliers (free-flyer GuSTO, third iteration of every instance: found by the
// teacher-forced test).  Swept on the host build over the teacher-forced goldens (tools/conic_reg_sweep.py): 1e-10 halves the
// refinement steps -- quadrotor GuSTO 6.4 -> 2.7 per iteration, free-flyer GuSTO 7.9 -> 3.0, Starship N = 100 3.2 -> 2.3 -- with
// every optimal value within 9e-8; 1e-11 starts to lose factorisations.)  Programs with FREE variables (equality-constrained only: the LCvx
// style guess programs) have pivots of exactly +-reg on them and intermediate magnitudes 1/reg: 1e-6 keeps those within
// what iterative refinement against the unregularised matrix repairs (swept on the Starship descent programs).
// Fewer cone rows than variables (m < n): P + Gt'Gt cannot have full rank from the cone rows alone, its small pivots rest on the
// regularisation again -- 1e-8 as before (tests/test_conic_cpu.py: a random LP with n = 12, m = 7, p = 2 loses its factorisation at 1e-10).
// Pure LPs (no second-order / exponential cone, no quadratic cost: the Starship programs) are degenerate and stay at 1e-8: the N = 11
// PTR program ends ALMOST_OPTIMAL in the nested order at 1e-10; 1e-9 would save a quarter of the refinement steps on the N = 100 SCvx
// programs (3.2 -> 2.3 per iteration, all 30 OPTIMAL, tools/conic_reg_sweep_starship.py) but the two orders then part by two
// iterations on one of six successive programs (tests/test_template_cpu.py asserts +-1).  With the safety nets of Opts::fine the host agrees
// again, but on the DEVICE the first subproblem of the 21 s Starship record then ended ALMOST_OPTIMAL 1.3e-5 off (gpurun_out/r05n): left alone.
// Programs with EXPONENTIAL cones (GuSTO pen = :softplus) stay at 1e-8 as well: on the device the second softplus subproblem of
// tests/test_gusto_gpu.py came out 60 % off at 1e-10 (gpurun_out/r05k).
inline double auto_reg(int n_free, int n, int m, bool pure_lp, bool has_exp = false)
{
    return n_free > 0 ? 1e-6 : ((m < n || pure_lp || has_exp) ? 1e-8 : 1e-10);
}
// nd_dense_factor / seen_ranks: see analyse_auto below (a dissection whose ranks are already in seen_ranks is not analysed
// again: the function returns early with nd_depth = -1).
inline Symbolic analyse(int n, int p, int m, int l, const std::vector<int>& q, const Csc& P, const Csc& A, const Csc& G,
                        const int* user_perm = nullptr, bool free_order = false, int ordering = ORDER_SEQUENTIAL,
                        double nd_dense_factor = 4.0, std::vector<std::vector<int>>* seen_ranks = nullptr)
{
    Symbolic S;
    S.n = n; S.p = p; S.m = m; S.l = l; S.q = q; S.nk = n + p + m;
    S.P = P; S.A = A; S.G = G;
    if (P.nrow != n || P.ncol != n) throw std::invalid_argument("P must be n x n");
    if (A.nrow != p || A.ncol != n) throw std::invalid_argument("A must be p x n");
    if (G.nrow != m || G.ncol != n) throw std::invalid_argument("G must be m x n");
    check_csc(P, "P", true); check_csc(A, "A", false); check_csc(G, "G", false);
    int tot = l;
    S.row_cone.assign(m, -1);
    for (size_t c = 0; c < q.size(); c++) {
        if (q[c] < 1) throw std::invalid_argument("cone dimension < 1");
        S.cone_off.push_back(tot);
        for (int r = 0; r < q[c]; r++) { if (tot + r < m) S.row_cone[tot + r] = (int)c; }
        tot += q[c];
    }
    if (l < 0 || tot != m) throw std::invalid_argument("l + sum(q) != m");

    // ---- Gt pattern: union of the rows of a cone per column ----
    S.Gt.nrow = m; S.Gt.ncol = n; S.Gt.p.assign(n + 1, 0);
    S.g2gt.assign(G.nnz(), -1);
    for (int c = 0; c < n; c++) {
        int last_cone = -1;
        for (int e = G.p[c]; e < G.p[c + 1]; e++) {
            const int r = G.i[e], cn = S.row_cone[r];
            if (cn < 0) {
                S.lp_gt.push_back((int)S.Gt.i.size()); S.lp_g.push_back(e);
                S.g2gt[e] = (int)S.Gt.i.size();
                S.Gt.i.push_back(r);
            } else {
                if (cn != last_cone) {
                    S.job_gt0.push_back((int)S.Gt.i.size()); S.job_cone.push_back(cn);
                    S.job_src_p.push_back((int)S.job_src_g.size());
                    for (int rr = 0; rr < q[cn]; rr++) S.Gt.i.push_back(S.cone_off[cn] + rr);
                    last_cone = cn;
                }
                S.job_src_row.push_back(r - S.cone_off[cn]); S.job_src_g.push_back(e);
                S.g2gt[e] = S.job_gt0.back() + (r - S.cone_off[cn]);
            }
        }
        S.Gt.p[c + 1] = (int)S.Gt.i.size();
    }
    S.job_src_p.push_back((int)S.job_src_g.size());
    S.Ar = csr_view(A); S.Gtr = csr_view(S.Gt);
    {   // symmetric expansion of P (row view incl. the mirrored entries)
        std::vector<std::vector<std::pair<int, int>>> rows(n);
        for (int c = 0; c < n; c++)
            for (int e = P.p[c]; e < P.p[c + 1]; e++) { rows[P.i[e]].push_back({c, e}); if (P.i[e] != c) rows[c].push_back({P.i[e], e}); }
        S.Pfull.p.assign(n + 1, 0);
        for (int r = 0; r < n; r++) {
            std::sort(rows[r].begin(), rows[r].end());
            for (auto& ce : rows[r]) { S.Pfull.j.push_back(ce.first); S.Pfull.pos.push_back(ce.second); }
            S.Pfull.p[r + 1] = (int)S.Pfull.j.size();
        }
    }

    {   // unified rows of Ktrue (see Symbolic::kk_p)
        const int nk_ = S.nk;
        S.kk_p.assign(nk_ + 1, 0);
        auto push = [&](int src, int idx, int col) { S.kk_src.push_back(src); S.kk_idx.push_back(idx); S.kk_col.push_back(col); };
        for (int i = 0; i < n; i++) {
            for (int t = S.Pfull.p[i]; t < S.Pfull.p[i + 1]; t++) push(1, S.Pfull.pos[t], S.Pfull.j[t]);
            for (int e = A.p[i]; e < A.p[i + 1]; e++) push(2, e, n + A.i[e]);
            for (int e = S.Gt.p[i]; e < S.Gt.p[i + 1]; e++) push(3, e, n + p + S.Gt.i[e]);
            S.kk_p[i + 1] = (int)S.kk_src.size();
        }
        for (int r = 0; r < p; r++) {
            for (int t = S.Ar.p[r]; t < S.Ar.p[r + 1]; t++) push(2, S.Ar.pos[t], S.Ar.j[t]);
            S.kk_p[n + r + 1] = (int)S.kk_src.size();
        }
        for (int r = 0; r < m; r++) {
            for (int t = S.Gtr.p[r]; t < S.Gtr.p[r + 1]; t++) push(3, S.Gtr.pos[t], S.Gtr.j[t]);
            S.kk_p[n + p + r + 1] = (int)S.kk_src.size();
        }
        for (int r = 0; r < nk_; r++) if (S.kk_p[r + 1] - S.kk_p[r] > Symbolic::KK_LONG) S.kk_long.push_back(r);
    }

    // ---- KKT adjacency (old numbering: x 0..n-1, y n..n+p-1, z n+p..) ----
    const int nk = S.nk;
    std::vector<std::vector<int>> adj(nk);
    auto edge = [&](int a, int b) { if (a != b) { adj[a].push_back(b); adj[b].push_back(a); } };
    for (int c = 0; c < n; c++) {
        for (int e = P.p[c]; e < P.p[c + 1]; e++) edge(P.i[e], c);
        for (int e = A.p[c]; e < A.p[c + 1]; e++) edge(n + A.i[e], c);
        for (int e = S.Gt.p[c]; e < S.Gt.p[c + 1]; e++) edge(n + p + S.Gt.i[e], c);
    }
    for (auto& v : adj) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
    if (user_perm) S.perm.assign(user_perm, user_perm + nk);
    else if (free_order) S.perm = min_degree(nk, adj);
    else {
        std::vector<int> cls(nk);
        for (int v = 0; v < nk; v++) cls[v] = v < n ? 1 : (v < n + p ? 2 : 0);
        std::vector<int> rank;
        // SCP_CONIC_ND_LEAF: node blocks per undissected leaf (tuning aid; 1 = dissect down to single nodes)
        const char* leaf_env = std::getenv("SCP_CONIC_ND_LEAF");
        if (ordering == ORDER_NESTED) {
            S.nd_depth = nd_ranks(n, p, adj, rank, leaf_env ? std::max(1, std::atoi(leaf_env)) : 1, nd_dense_factor);
            if (seen_ranks) {
                if (S.nd_depth == 0 || std::find(seen_ranks->begin(), seen_ranks->end(), rank) != seen_ranks->end()) { S.nd_depth = -1; return S; }
                seen_ranks->push_back(rank);
            }
        }
        S.perm = min_degree(nk, adj, &cls, true, S.nd_depth > 0 ? &rank : nullptr);
    }
    S.iperm.assign(nk, -1);
    for (int k = 0; k < nk; k++) {
        if (S.perm[k] < 0 || S.perm[k] >= nk || S.iperm[S.perm[k]] != -1) throw std::invalid_argument("ordering is not a permutation");
        S.iperm[S.perm[k]] = k;
    }

    // ---- symbolic factorisation: column patterns of L by the elimination-tree merge ----
    // cols[j] = sorted rows i > j of L(:,j); L(:,j) = pattern(K(j+1:,j)) U union over children c of (L(:,c) \ {j})
    std::vector<std::vector<int>> cols(nk), kids(nk);
    for (int j = 0; j < nk; j++) {
        std::vector<int>& cj = cols[j];
        for (int w : adj[S.perm[j]]) { const int iw = S.iperm[w]; if (iw > j) cj.push_back(iw); }
        for (int c : kids[j]) for (int r : cols[c]) if (r > j) cj.push_back(r);
        std::sort(cj.begin(), cj.end()); cj.erase(std::unique(cj.begin(), cj.end()), cj.end());
        if (!cj.empty()) kids[cj[0]].push_back(j);   // parent = first off-diagonal row
    }
    S.Lp.assign(nk + 1, 0);
    for (int j = 0; j < nk; j++) S.Lp[j + 1] = S.Lp[j] + (int)cols[j].size();
    S.Li.resize(S.Lp[nk]);
    for (int j = 0; j < nk; j++) std::copy(cols[j].begin(), cols[j].end(), S.Li.begin() + S.Lp[j]);
    const int nnzL = S.Lp[nk];

    // ---- sources of the KKT values ----
    S.l_src.assign(nnzL, SRC_NONE); S.l_src_idx.assign(nnzL, 0);
    S.d_src.assign(nk, SRC_NONE); S.d_src_idx.assign(nk, 0); S.d_kind.assign(nk, 0);
    for (int j = 0; j < nk; j++) { const int o = S.perm[j]; S.d_kind[j] = o < n ? 0 : (o < n + p ? 1 : 2); }
    auto find_l = [&](int a, int b) {   // entry (max, min) of L in permuted numbering
        const int i = std::max(a, b), j = std::min(a, b);
        auto it = std::lower_bound(S.Li.begin() + S.Lp[j], S.Li.begin() + S.Lp[j + 1], i);
        if (it == S.Li.begin() + S.Lp[j + 1] || *it != i) throw std::logic_error("KKT entry missing from L");
        return (int)(it - S.Li.begin());
    };
    for (int c = 0; c < n; c++) {
        for (int e = P.p[c]; e < P.p[c + 1]; e++) {
            if (P.i[e] == c) { S.d_src[S.iperm[c]] = SRC_P; S.d_src_idx[S.iperm[c]] = e; }
            else { const int t = find_l(S.iperm[P.i[e]], S.iperm[c]); S.l_src[t] = SRC_P; S.l_src_idx[t] = e; }
        }
        for (int e = A.p[c]; e < A.p[c + 1]; e++) { const int t = find_l(S.iperm[n + A.i[e]], S.iperm[c]); S.l_src[t] = SRC_A; S.l_src_idx[t] = e; }
        for (int e = S.Gt.p[c]; e < S.Gt.p[c + 1]; e++) { const int t = find_l(S.iperm[n + p + S.Gt.i[e]], S.iperm[c]); S.l_src[t] = SRC_GT; S.l_src_idx[t] = e; }
    }

    // ---- row lists of L and the pair schedule ----
    S.row_p.assign(nk + 1, 0);
    for (int e = 0; e < nnzL; e++) S.row_p[S.Li[e] + 1]++;
    for (int r = 0; r < nk; r++) S.row_p[r + 1] += S.row_p[r];
    S.row_k.resize(nnzL); S.row_pos.resize(nnzL);
    {
        std::vector<int> w(S.row_p.begin(), S.row_p.end() - 1);
        for (int j = 0; j < nk; j++)
            for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) { const int t = w[S.Li[e]]++; S.row_k[t] = j; S.row_pos[t] = e; }
    }
    S.pair_p.assign(nnzL + 1, 0);
    int64_t flops = 0;
    for (int pass = 0; pass < 2; pass++) {
        int64_t cnt = 0;
        for (int j = 0; j < nk; j++) {
            const int rj0 = S.row_p[j], rj1 = S.row_p[j + 1];
            for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) {
                const int i = S.Li[e];
                int a = S.row_p[i], b = rj0;
                const int a1 = S.row_p[i + 1];
                if (pass == 1) S.pair_p[e] = cnt;
                while (a < a1 && b < rj1) {
                    const int ka = S.row_k[a], kb = S.row_k[b];
                    if (ka >= j) break;
                    if (ka == kb) { if (pass == 1) { S.pair_a[cnt] = S.row_pos[a]; S.pair_b[cnt] = S.row_pos[b]; } cnt++; a++; b++; }
                    else if (ka < kb) a++;
                    else b++;
                }
            }
        }
        if (pass == 0) { S.pair_a.resize(cnt); S.pair_b.resize(cnt); flops = cnt; }
        else S.pair_p[nnzL] = cnt;
    }
    S.flops = flops + nnzL;

    // ---- free variables ----
    {
        std::vector<char> solid(n, 0);
        for (int c = 0; c < n; c++) {
            if (G.p[c + 1] > G.p[c]) solid[c] = 1;
            for (int e = P.p[c]; e < P.p[c + 1]; e++) if (P.i[e] == c) solid[c] = 1;
        }
        S.n_free = 0;
        for (int c = 0; c < n; c++) S.n_free += solid[c] ? 0 : 1;
    }

    // ---- level sets ----
    {
        std::vector<int> lev(nk, 0), rlev(nk, 0);
        int nlev = 0, nrlev = 0;
        for (int j = 0; j < nk; j++) {
            int lv = 0;
            for (int t = S.row_p[j]; t < S.row_p[j + 1]; t++) lv = std::max(lv, lev[S.row_k[t]] + 1);
            lev[j] = lv; nlev = std::max(nlev, lv + 1);
        }
        for (int j = nk - 1; j >= 0; j--) {
            int lv = 0;
            for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) lv = std::max(lv, rlev[S.Li[e]] + 1);
            rlev[j] = lv; nrlev = std::max(nrlev, lv + 1);
        }
        auto bucket = [&](const std::vector<int>& lv, int nl, std::vector<int>& ptr, std::vector<int>& items) {
            ptr.assign(nl + 1, 0);
            for (int j = 0; j < nk; j++) ptr[lv[j] + 1]++;
            for (int q2 = 0; q2 < nl; q2++) ptr[q2 + 1] += ptr[q2];
            items.resize(nk);
            std::vector<int> w(ptr.begin(), ptr.end() - 1);
            for (int j = 0; j < nk; j++) items[w[lv[j]]++] = j;
        };
        bucket(lev, nlev, S.lev_p, S.lev_cols);
        bucket(rlev, nrlev, S.rlev_p, S.rlev_cols);
        S.ent_col.resize(nnzL);
        for (int j = 0; j < nk; j++) for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) S.ent_col[e] = j;
        S.lev_ent_p.assign(nlev + 1, 0);
        S.lev_ent.clear(); S.lev_ent.reserve(nnzL);
        for (int q2 = 0; q2 < nlev; q2++) {
            for (int t = S.lev_p[q2]; t < S.lev_p[q2 + 1]; t++) {
                const int j = S.lev_cols[t];
                for (int e = S.Lp[j]; e < S.Lp[j + 1]; e++) S.lev_ent.push_back(e);
            }
            S.lev_ent_p[q2 + 1] = (int)S.lev_ent.size();
        }
        // ---- long rows / pair lists -> chunks (see Symbolic) ----
        // SCP_CONIC_LONG_ITEM: threshold above which an item is cut (tuning aid; default Symbolic::LONG_ITEM)
        const char* li_env = std::getenv("SCP_CONIC_LONG_ITEM");
        const int long_item = li_env ? std::max(8, std::atoi(li_env)) : Symbolic::LONG_ITEM;
        auto chunk_len = [&](int64_t len) { return std::max<int64_t>(std::min(32, long_item), (int64_t)std::ceil(std::sqrt((double)len))); };
        S.lev_nshort.assign(nlev, 0); S.lev_ent_nshort.assign(nlev, 0);
        S.rchunk_p.assign(nlev + 1, 0); S.echunk_p.assign(nlev + 1, 0);
        S.col_c0.assign(nk, 0); S.col_c1.assign(nk, 0);
        S.ent_c0.assign(nnzL, 0); S.ent_c1.assign(nnzL, 0);
        S.max_chunks = 1;
        for (int q2 = 0; q2 < nlev; q2++) {
            auto row_len = [&](int j) { return S.row_p[j + 1] - S.row_p[j]; };
            auto pair_len = [&](int e) { return S.pair_p[e + 1] - S.pair_p[e]; };
            auto c_first = S.lev_cols.begin() + S.lev_p[q2], c_last = S.lev_cols.begin() + S.lev_p[q2 + 1];
            auto c_mid = std::stable_partition(c_first, c_last, [&](int j) { return row_len(j) <= long_item; });
            S.lev_nshort[q2] = (int)(c_mid - c_first);
            for (auto it = c_mid; it != c_last; ++it) {
                const int t = (int)(it - S.lev_cols.begin()), j = *it;
                const int cl = (int)chunk_len(row_len(j));
                S.col_c0[t] = (int)S.rchunk_r0.size();
                for (int r = S.row_p[j]; r < S.row_p[j + 1]; r += cl) { S.rchunk_r0.push_back(r); S.rchunk_r1.push_back(std::min(r + cl, S.row_p[j + 1])); }
                S.col_c1[t] = (int)S.rchunk_r0.size();
            }
            S.rchunk_p[q2 + 1] = (int)S.rchunk_r0.size();
            auto e_first = S.lev_ent.begin() + S.lev_ent_p[q2], e_last = S.lev_ent.begin() + S.lev_ent_p[q2 + 1];
            auto e_mid = std::stable_partition(e_first, e_last, [&](int e) { return pair_len(e) <= long_item; });
            S.lev_ent_nshort[q2] = (int)(e_mid - e_first);
            for (auto it = e_mid; it != e_last; ++it) {
                const int t = (int)(it - S.lev_ent.begin()), e = *it;
                const int64_t cl = chunk_len(pair_len(e));
                S.ent_c0[t] = (int)S.echunk_q0.size();
                for (int64_t qq = S.pair_p[e]; qq < S.pair_p[e + 1]; qq += cl) { S.echunk_q0.push_back(qq); S.echunk_q1.push_back(std::min<int64_t>(qq + cl, S.pair_p[e + 1])); }
                S.ent_c1[t] = (int)S.echunk_q0.size();
            }
            S.echunk_p[q2 + 1] = (int)S.echunk_q0.size();
            S.max_chunks = std::max(S.max_chunks, std::max(S.rchunk_p[q2 + 1] - S.rchunk_p[q2], S.echunk_p[q2 + 1] - S.echunk_p[q2]));
        }
    }
    return S;
}

// Cost of one factorisation + substitution sweep of a schedule on the level-scheduled kernel, in units of one elimination
// level: a level costs its two workgroup barriers however little work it holds, the multiply-adds are shared by the
// `workers` lanes a problem has (16 ... 1 024, conic_api.hip: launch geometry by batch size).  Two measurements on MI355X
// fix the exchange rate.  Quadrotor GuSTO program N = 30 at 1 024 problems, 256 workers each (seconds per launch of 17
// iterations: 49 levels / 62 k multiply-adds 0.25, 204 / 48 k 0.48, 192 / 947 k 1.39): a level is worth 6 multiply-adds
// per worker.  Literal rocket program N = 100 at 16 384 problems, 16 workers each (68 levels / 419 k: 6.85 s, 713 / 355 k:
// 8.24 s): 15 per worker.  12 ranks every measured pair correctly.  Large batches (few workers per problem) lean towards the
// order with fewer multiply-adds, small batches towards the one with fewer levels.
inline double schedule_cost(const Symbolic& S, int workers)
{
    return (double)((long)S.lev_p.size() - 1) + (double)S.flops / (12.0 * (double)(workers > 0 ? workers : 1));
}

// The ordering the solver uses when the caller leaves it open: the cheapest (schedule_cost) of the sequential order and
// the nested dissections found with a few thresholds for "globally coupled" vertices.  nd_ranks drops the vertices whose
// degree exceeds dense_factor x median before it looks for the chain; which variables that catches depends on the program
// (a trust-region epigraph that touches every node has degree ~2 N: 61 at N = 30, below 4 x median when the median
// block is large -- it then ties the whole horizon into one component and the dissection finds no chain, or a useless one
// of depth 2 whose separators fill in: measured on the slack-free GuSTO program, 20 x the multiply-adds at the same
// level count).  Trying 4, 2 and 1.25 and pricing the results costs a few symbolic analyses at create time.
// best_is_nested: the winner is a dissection (the caller keeps the sequential schedule as its fallback).
inline Symbolic analyse_auto(int n, int p, int m, int l, const std::vector<int>& q, const Csc& P, const Csc& A, const Csc& G,
                             int workers, bool allow_sequential, Symbolic* sequential_out, bool* best_is_nested)
{
    // the candidate orders are independent analyses of the same pattern: one host thread each (create of the free-flyer N = 200
    // template: four analyses of a KKT pattern with 3e5 factor entries -- the wall time of `create` is the slowest one, not the sum)
    auto fut_seq = std::async(std::launch::async, [&] { return analyse(n, p, m, l, q, P, A, G, nullptr, false, ORDER_SEQUENTIAL); });
    const double factors[3] = {4.0, 2.0, 1.25};
    std::future<Symbolic> fut_nd[3];
    for (int i = 0; i < 3; i++)
        fut_nd[i] = std::async(std::launch::async, [&, i] { return analyse(n, p, m, l, q, P, A, G, nullptr, false, ORDER_NESTED, factors[i], nullptr); });
    Symbolic seq = fut_seq.get();
    Symbolic best;
    bool have = false;
    for (int i = 0; i < 3; i++) {
        Symbolic S = fut_nd[i].get();
        if (S.nd_depth <= 0) continue;
        if (!have || schedule_cost(S, workers) < schedule_cost(best, workers)) { best = std::move(S); have = true; }    // (ties keep the earlier factor, as before)
    }
    const bool nested = have && (!allow_sequential || schedule_cost(best, workers) < schedule_cost(seq, workers));
    if (best_is_nested) *best_is_nested = nested;
    if (sequential_out) *sequential_out = seq;
    return nested ? best : seq;
}

}  // namespace conic
}  // namespace scp
