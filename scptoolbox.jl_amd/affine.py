"""Symbolic-affine assembly of conic programs: the reference's per-iteration JuMP formulation, hoisted out of the loop.

The reference builds a NEW JuMP model every SCP iteration (`Subproblem(pbm, iter, ref)`, src/solvers/ptr.jl:213-293;
`@add_constraint` evaluates the user expression on AffExprs, src/parser/constraint.jl:120-141).  Every coefficient it
writes is an AFFINE function of a small set of per-problem, per-iteration numbers -- the discretised dynamics
`ref.dyn.{A,B,F,r,E}`, the reference trajectory, the Jacobians of the non-convex constraints and boundary conditions,
the trust-region radius -- with constant coefficients (scaling matrices, weights).  This module records that map once:

    value[slot] = const[slot] + sum_t coef[t] * src[src_index[t]]

for the value arrays (c, b, h, Gx, Ax, Px) of the standard form of include/scp_conic.h, on a sparsity pattern that does
not depend on the numbers.  The device then re-fills the values of a whole Monte-Carlo batch with one gather kernel
(csrc/scp_generic.hip) instead of re-running a modelling layer per problem and iteration.

`Aff` is a dense array of affine scalars; `ConicAssembler` mirrors the subset of the reference's DSL the SCP solvers
use (ZERO / NONPOS / SOC / L1 / LINF cones, src/parser/cone.jl:36-47, with MOI's NormOne / NormInfinity bridges
spelled out, linear + diagonal-quadratic cost).
"""
import numpy as np
import scipy.sparse as sp


class Sources:
    """Named segments of the per-problem source vector (each segment in Julia / column-major order)."""

    def __init__(self):
        self.segs = {}
        self.n = 0

    def add(self, name, shape):
        shape = tuple(int(s) for s in np.atleast_1d(shape))
        if name in self.segs:
            raise KeyError(name)
        self.segs[name] = (self.n, shape)
        self.n += int(np.prod(shape))
        return self.segs[name][0]

    def ref(self, name):
        """Aff array of shape `shape` whose element (i, j, ...) is the source value itself."""
        off, shape = self.segs[name]
        size = int(np.prod(shape))
        src = off + np.arange(size).reshape(shape, order="F")      # column-major position of (i, j, ...)
        pos = np.arange(size)                                      # C-order flat position inside the Aff
        return Aff(np.zeros(shape), pos, src.reshape(-1), np.ones(size))


class Aff:
    """Array of affine scalars  c0 + sum coef * src[.]  (terms: flat C-order position, source index, coefficient)."""
    __array_priority__ = 100

    def __init__(self, c0, pos=None, src=None, coef=None):
        self.c0 = np.array(c0, dtype=float)
        z = np.zeros(0, dtype=np.int64)
        self.pos = z if pos is None else np.asarray(pos, np.int64).reshape(-1)
        self.src = z if src is None else np.asarray(src, np.int64).reshape(-1)
        self.coef = np.zeros(0) if coef is None else np.asarray(coef, float).reshape(-1)

    @property
    def shape(self):
        return self.c0.shape

    @staticmethod
    def lift(a):
        return a if isinstance(a, Aff) else Aff(np.asarray(a, float))

    def __neg__(self):
        return Aff(-self.c0, self.pos, self.src, -self.coef)

    def __add__(self, o):
        o = Aff.lift(o)
        if o.shape != self.shape:
            if o.pos.size == 0:
                o = Aff(np.broadcast_to(o.c0, self.shape))
            elif self.pos.size == 0:
                return o + self
            else:
                raise ValueError("shape mismatch %s vs %s" % (self.shape, o.shape))
        return Aff(self.c0 + o.c0, np.concatenate([self.pos, o.pos]), np.concatenate([self.src, o.src]),
                   np.concatenate([self.coef, o.coef]))

    __radd__ = __add__

    def __sub__(self, o):
        return self + (-Aff.lift(o))

    def __rsub__(self, o):
        return Aff.lift(o) + (-self)

    def __mul__(self, k):
        """element-wise product with a CONSTANT (broadcast to this shape)."""
        k = np.broadcast_to(np.asarray(k, float), self.shape)
        return Aff(self.c0 * k, self.pos, self.src, self.coef * k.reshape(-1)[self.pos])

    __rmul__ = __mul__

    def __matmul__(self, v):
        """(m x n affine matrix) @ (constant n-vector) -> affine m-vector."""
        v = np.asarray(v, float)
        assert self.c0.ndim == 2 and v.shape == (self.shape[1],)
        n = self.shape[1]
        return Aff(self.c0 @ v, self.pos // n, self.src, self.coef * v[self.pos % n])

    def __getitem__(self, key):
        idx = np.arange(self.c0.size).reshape(self.shape)[key]
        lut = np.full(self.c0.size, -1, np.int64)
        lut[np.asarray(idx).reshape(-1)] = np.arange(np.asarray(idx).size)
        new = lut[self.pos]
        keep = new >= 0
        return Aff(self.c0[key], new[keep], self.src[keep], self.coef[keep])

    def reshape(self, *shape):
        return Aff(self.c0.reshape(*shape), self.pos, self.src, self.coef)

    @staticmethod
    def vstack(blocks):
        """stack 2-D blocks (Aff or ndarray) vertically."""
        blocks = [Aff.lift(np.atleast_2d(b)) if not isinstance(b, Aff) else b for b in blocks]
        ncol = blocks[0].shape[1]
        c0 = np.vstack([b.c0 for b in blocks])
        pos, src, coef, off = [], [], [], 0
        for b in blocks:
            assert b.c0.ndim == 2 and b.shape[1] == ncol
            pos.append(b.pos + off * ncol); src.append(b.src); coef.append(b.coef)
            off += b.shape[0]
        return Aff(c0, np.concatenate(pos), np.concatenate(src), np.concatenate(coef))

    def evaluate(self, src):
        """numerical value for one source vector (host-side check of the device gather)."""
        out = self.c0.reshape(-1).copy()
        np.add.at(out, self.pos, self.coef * np.asarray(src, float)[self.src])
        return out.reshape(self.shape)


class AffineMap:
    """value = const + T @ src for one value array (CSR over slots)."""

    def __init__(self, const, ptr, src, coef):
        self.const = np.ascontiguousarray(const, np.float64)
        self.ptr = np.ascontiguousarray(ptr, np.int32)
        self.src = np.ascontiguousarray(src, np.int32)
        self.coef = np.ascontiguousarray(coef, np.float64)

    @property
    def size(self):
        return self.const.size

    def evaluate(self, src):
        out = self.const.copy()
        if self.src.size:
            rows = np.repeat(np.arange(self.size), np.diff(self.ptr))
            np.add.at(out, rows, self.coef * np.asarray(src, float)[self.src])
        return out

    @staticmethod
    def from_terms(const, slot, src, coef):
        const = np.asarray(const, float)
        order = np.argsort(slot, kind="stable")
        slot, src, coef = np.asarray(slot)[order], np.asarray(src)[order], np.asarray(coef)[order]
        ptr = np.zeros(const.size + 1, np.int64)
        np.add.at(ptr, slot + 1, 1)
        return AffineMap(const, np.cumsum(ptr), src, coef)


class ConicTemplate:
    """Pattern + affine value maps of one family of conic programs (include/scp_conic.h standard form)."""

    def __init__(self, n, l, q, G, A, P, maps, variables, nsrc):
        self.n, self.l, self.q = n, l, list(q)
        self.m = l + sum(abs(v) for v in self.q)
        self.p = A.shape[0]
        self.G, self.A, self.P = G, A, P          # scipy CSC patterns (data = constant parts)
        self.maps = maps                          # dict: c, b, h, Gx, Ax, Px -> AffineMap
        self.variables = variables                # name -> index array into x
        self.nsrc = nsrc

    def values(self, src):
        """(c, b, h, Gx, Ax, Px) for one source vector."""
        return {k: m.evaluate(src) for k, m in self.maps.items()}


class ConicAssembler:
    """Flat-variable conic program assembler with affine coefficients."""

    def __init__(self, sources):
        self.sources = sources
        self.n = 0
        self.variables = {}
        self.eq, self.nonpos, self.soc, self.exp = [], [], [], []
        self.c_terms, self.P_terms = [], []

    def var(self, n, name=None):
        idx = np.arange(self.n, self.n + n)
        self.n += n
        if name is not None:
            self.variables[name] = idx if name not in self.variables else np.concatenate([self.variables[name], idx])
        return idx

    @staticmethod
    def _block(terms, const):
        const = Aff.lift(const)
        m = const.shape[0]
        out = []
        for idx, M in terms:
            M = Aff.lift(M)
            if M.c0.ndim == 1:
                M = M.reshape(1, -1)
            assert M.shape == (m, len(idx)), (M.shape, m, len(idx))
            out.append((np.asarray(idx), M))
        return out, const

    def add_zero(self, terms, const):          # expr == 0      (ZERO cone)
        self.eq.append(self._block(terms, const))

    def add_nonpos(self, terms, const):        # expr <= 0      (NONPOS cone)
        self.nonpos.append(self._block(terms, const))

    def add_soc(self, terms, const):           # expr in Q      (SOC cone: expr[0] >= ||expr[1:]||)
        self.soc.append(self._block(terms, const))

    def add_exp(self, terms, const):           # expr = (x, y, w) in EXP: y exp(x / y) <= w, y > 0  (src/parser/cone.jl:45)
        blk = self._block(terms, const)
        assert blk[1].shape[0] == 3
        self.exp.append(blk)

    def add_linf(self, t_idx, terms, const):
        """t >= ||expr||_inf (LINF cone; MOI NormInfinity bridge: expr - t <= 0, -expr - t <= 0)."""
        const = Aff.lift(const)
        ones = np.ones((const.shape[0], 1))
        self.add_nonpos(list(terms) + [(t_idx, -ones)], const)
        self.add_nonpos([(i, -Aff.lift(M)) for i, M in terms] + [(t_idx, -ones)], -const)

    def add_l1(self, t_idx, terms, const, name=None):
        """t >= ||expr||_1 (L1 cone; MOI NormOne bridge: |expr_i| <= y_i, sum y <= t)."""
        const = Aff.lift(const)
        m = const.shape[0]
        y = self.var(m, name)
        I = np.eye(m)
        self.add_nonpos(list(terms) + [(y, -I)], const)
        self.add_nonpos([(i, -Aff.lift(M)) for i, M in terms] + [(y, -I)], -const)
        self.add_nonpos([(y, np.ones((1, m))), (t_idx, -np.ones((1, 1)))], np.zeros(1))

    def add_cost_lin(self, idx, w):
        self.c_terms.append((np.atleast_1d(idx), Aff.lift(np.atleast_1d(w)) if not isinstance(w, Aff) else w))

    def add_cost_quad_diag(self, idx, w):
        """+ sum_i w_i x_i^2  ->  P_ii += 2 w_i  (w: constants or affine scalars, e.g. GuSTO's lambda-weighted penalties)."""
        w = w if isinstance(w, Aff) else Aff(np.atleast_1d(np.asarray(w, float)))
        self.P_terms.append((np.atleast_1d(idx), w * 2.0))

    # ------------------------------------------------------------------------------------------------------
    def _stack(self, blocks, sign):
        """-> (rows, cols, c0, term lists) of the stacked matrix and the stacked constant Aff."""
        R, C, V0, TS, TC, TE = [], [], [], [], [], []      # entry rows / cols / const ; term src / coef / entry id
        consts = []
        off = 0
        ne = 0
        for terms, const in blocks:
            m = const.shape[0]
            for idx, M in terms:
                ncol = len(idx)
                has = (M.c0 != 0.0).reshape(-1)
                has[M.pos] = True
                flat = np.nonzero(has)[0]
                eid = np.full(M.c0.size, -1, np.int64)
                eid[flat] = ne + np.arange(flat.size)
                R.append(off + flat // ncol); C.append(idx[flat % ncol]); V0.append(sign * M.c0.reshape(-1)[flat])
                TS.append(M.src); TC.append(sign * M.coef); TE.append(eid[M.pos])
                ne += flat.size
            consts.append(const)
            off += m
        cat = lambda L, dt: np.concatenate(L).astype(dt) if L else np.zeros(0, dt)
        return off, cat(R, np.int64), cat(C, np.int64), cat(V0, float), cat(TS, np.int64), cat(TC, float), cat(TE, np.int64), consts

    @staticmethod
    def _vec_map(consts, sign, nsrc):
        c0, slot, src, coef = [], [], [], []
        off = 0
        for a in consts:
            c0.append(sign * a.c0.reshape(-1)); slot.append(a.pos + off); src.append(a.src); coef.append(sign * a.coef)
            off += a.c0.size
        cat = lambda L, dt: np.concatenate(L).astype(dt) if L else np.zeros(0, dt)
        return AffineMap.from_terms(cat(c0, float), cat(slot, np.int64), cat(src, np.int64), cat(coef, float))

    def _matrix(self, nrow, R, C, V0, TS, TC, TE):
        """merge duplicate (row, col) entries, order CSC; returns (csc with const data, AffineMap of the values)."""
        key = C * max(nrow, 1) + R
        uniq, inv = np.unique(key, return_inverse=True)           # sorted by column then row = CSC order
        v0 = np.zeros(uniq.size)
        np.add.at(v0, inv, V0)
        rows, cols = uniq % max(nrow, 1), uniq // max(nrow, 1)
        indptr = np.zeros(self.n + 1, np.int64)
        np.add.at(indptr, cols + 1, 1)
        M = sp.csc_matrix((v0, rows.astype(np.int32), np.cumsum(indptr).astype(np.int32)), shape=(nrow, self.n))
        return M, AffineMap.from_terms(v0, inv[TE] if TE.size else np.zeros(0, np.int64), TS, TC)

    def finalize(self):
        nsrc = self.sources.n
        # A x = b  (blocks are expr = A x + a0 == 0)
        p, R, C, V0, TS, TC, TE, consts = self._stack(self.eq, 1.0)
        A, mapA = self._matrix(p, R, C, V0, TS, TC, TE)
        mapb = self._vec_map(consts, -1.0, nsrc)
        # G x + s = h : NONPOS rows (expr <= 0 -> G = M, h = -g0), then SOC blocks (expr in Q -> G = -M, h = m0)
        l, R1, C1, V1, TS1, TC1, TE1, c1 = self._stack(self.nonpos, 1.0)
        # (exponential cones after the second-order cones; q = -3 marks one, include/scp_conic.h)
        ms, R2, C2, V2, TS2, TC2, TE2, c2 = self._stack(self.soc + self.exp, -1.0)
        G, mapG = self._matrix(l + ms, np.concatenate([R1, R2 + l]), np.concatenate([C1, C2]), np.concatenate([V1, V2]),
                               np.concatenate([TS1, TS2]), np.concatenate([TC1, TC2]),
                               np.concatenate([TE1, TE2 + (R1.size)]))
        h1, h2 = self._vec_map(c1, -1.0, nsrc), self._vec_map(c2, 1.0, nsrc)
        maph = AffineMap(np.concatenate([h1.const, h2.const]), np.concatenate([h1.ptr, h2.ptr[1:] + h1.ptr[-1]]),
                         np.concatenate([h1.src, h2.src]), np.concatenate([h1.coef, h2.coef]))
        q = [const.shape[0] for _, const in self.soc] + [-3] * len(self.exp)
        # cost
        cv = Aff(np.zeros(self.n))
        for idx, w in self.c_terms:
            w = Aff.lift(w)
            cv = cv + Aff(np.bincount(idx, weights=w.c0.reshape(-1), minlength=self.n), idx[w.pos], w.src, w.coef)
        mapc = AffineMap.from_terms(cv.c0, cv.pos, cv.src, cv.coef)
        Pv = Aff(np.zeros(self.n))
        for idx, w in self.P_terms:
            Pv = Pv + Aff(np.bincount(idx, weights=w.c0.reshape(-1), minlength=self.n), idx[w.pos], w.src, w.coef)
        has = Pv.c0 != 0.0
        has[Pv.pos] = True
        nzp = np.nonzero(has)[0]
        lut = np.full(self.n, -1, np.int64); lut[nzp] = np.arange(nzp.size)
        P = sp.csc_matrix((Pv.c0[nzp], (nzp, nzp)), shape=(self.n, self.n))
        mapP = AffineMap.from_terms(Pv.c0[nzp], lut[Pv.pos], Pv.src, Pv.coef)
        if self.row_scaling:
            # Static row equilibration (what ECOS's default equilibration does to the reference's programs): the formulation
            # works in scaled VARIABLES but physical ROWS, e.g. a thrust bound has coefficients and a right-hand side of
            # 6e6 next to trust-region rows of 1 -- the relative residual tests then see only the large rows and the KKT
            # pivots span 13 decades.  Every row is multiplied by max|constant coefficient|^(-1/2) (one factor per
            # second-order cone); the primal solution and the cost are unchanged, only the multipliers are rescaled.
            self._scale_rows(A, mapA, mapb, [np.array([i]) for i in range(p)])
            groups = [np.array([i]) for i in range(l)]
            o = l
            for d in q:      # one factor per second-order cone keeps the cone a cone; exponential cones are left alone: a common
                if d > 0:    # factor would keep them cones too, but their scaling (dual barrier Hessian) is NOT invariant
                    groups.append(np.arange(o, o + d))      # under it and the solver starts them on the central ray of the
                o += abs(d)                                 # rows as written (measured: scaled rows -> iteration limit)
            self._scale_rows(G, mapG, maph, groups)
        maps = dict(c=mapc, b=mapb, h=maph, Gx=mapG, Ax=mapA, Px=mapP)
        return ConicTemplate(self.n, l, q, G, A, P, maps, dict(self.variables), nsrc)

    row_scaling = True
    row_scaling_power = 0.5

    def _scale_rows(self, M, mapM, mapr, groups):
        if M.shape[0] == 0:
            return
        rows = M.indices
        mag = np.zeros(M.shape[0])
        np.maximum.at(mag, rows, np.abs(mapM.const))
        e = np.ones(M.shape[0])
        for g in groups:
            r = mag[g].max()
            if r > 0.0:
                e[g] = r ** (-self.row_scaling_power)
        per_entry = e[rows]
        mapM.const *= per_entry
        mapM.coef *= np.repeat(per_entry, np.diff(mapM.ptr))
        M.data[:] = mapM.const
        mapr.const *= e
        mapr.coef *= np.repeat(e, np.diff(mapr.ptr))
