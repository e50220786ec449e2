"""Exact referee for the decisions of the global and local quantisers (test infrastructure).

On content whose decisions are exact ties in exact arithmetic -- perfect gradients (evenly spaced collinear colours), a handful of
distinct colours, one flat colour -- the reference's own choices hinge on the rounding of its SEQUENTIAL f64 sums
(lib/src/quantize/local.c:118-134, cells.c:82-112, math/pca.c:84-97, cluster.c:111-152), of an OpenBLAS `dgemv` whose kernel is
picked by CPU model (sort.c:43-56) and of LAPACK's `dsyev`; the reference has no machine-independent answer there.  Another
summation order (the HIP path's order-free sums, or the oracle's own sums taken back to front) gives another member of that tie
set: other cuts, other palette rows, another image.  This module turns "it is tie noise" into a checked statement.

Given the pixels (in the quantisation space), the weights and a SPLIT TRACE -- what a run decided: the global quantiser's axis,
covariance and cuts, and per committed split of the greedy loop the row, axis, covariance, bucket, member counts, distortions
(oracle: orc_last_split_trace; HIP path: patolette_amd_last_split_trace) -- `Referee.check` replays the run decision by
decision and verifies each one against EXACT arithmetic (Python integers: every f64 is m * 2^e) on the cluster's own pixels:

  covariance   the matrix handed to the eigen-solver equals the exact (weighted) covariance to within the envelope
  axis         = dsyev of that very matrix, bit for bit (the oracle's LAPACK restatement, pinned against real LAPACK)
  members      the traced axis + bucket give the traced member counts (sort.c:81-87; pixels within a rounding of a bucket
               border are allowed either side)
  cut          objective(bucket) >= max_b objective(b) - envelope                      (local.c:136-176)
  greedy step  benefit(row) >= benefit(every other frontier cluster) - envelope        (local.c:256-307)
  stop         benefit < 1e-16 (+- envelope) exactly when the run stopped              (local.c:365-370)
  GQ cuts      cost(cuts) <= optimum of the reference's DP + envelope; termination rule  (global.c:189-298)

envelope = 64 * n * 2^-53 * |value|: what n sequentially added, individually rounded terms can be off by (n * 2^-53 * sum |terms|)
with a factor 64 for the products, the division and the second operand.  A decision that is the exact optimum passes with margin
0; a decision inside the envelope is a proven TIE (recorded in the report); anything else is a VIOLATION: a bug, not noise.
`first_divergence` names the first decision two traces disagree on.

Only tests/ import this; it needs the oracle (eigen restatement) and numpy, no GPU.
"""
from fractions import Fraction

import numpy as np

EPS = 2.0 ** -53
ENV = 64.0
BUCKETS = 512
DELTA = 1e-16
GQ_MAX_K = 12


def _exact_ints(a):
    """f64 array -> (object array of Python ints X, S) with a[i] == X[i] / 2**S exactly."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    m, e = np.frexp(a)
    M = (m * 2.0 ** 53).astype(np.int64)
    ex = e.astype(np.int64) - 53
    nz = M != 0
    S = int(-ex[nz].min()) if nz.any() else 0
    S = max(S, 0)
    X = np.empty(a.shape[0], dtype=object)
    sh = ex + S
    for i in range(a.shape[0]):
        X[i] = (int(M[i]) << int(sh[i])) if M[i] else 0
    return X, S


class Report:
    def __init__(self):
        self.violations = []      # strings: decisions outside the envelope (bugs)
        self.ties = []            # dicts: decisions that are not the exact optimum but inside the envelope
        self.decisions = 0        # decisions checked
        self.ambiguous_members = 0

    @property
    def ok(self):
        return not self.violations

    def tie(self, kind, step, gap, env):
        self.ties.append(dict(kind=kind, step=step, gap=float(gap), envelope=float(env)))

    def summary(self):
        kinds = {}
        for t in self.ties:
            kinds[t["kind"]] = kinds.get(t["kind"], 0) + 1
        return "decisions %d, ties %s, violations %d%s" % (self.decisions, kinds, len(self.violations),
                                                          ("; first: " + self.violations[0]) if self.violations else "")


class _Cluster:
    __slots__ = ("idx", "SW", "S1", "S2", "SZ", "dist", "split_rec", "children", "benefit", "n")


class Referee:
    """colors: planar flat f64 (3n) in the quantisation space; weights: n f64 or None."""

    def __init__(self, flat, weights, n, eigen):
        self.n = n
        self.c = np.asarray(flat, dtype=np.float64).reshape(3, n)
        self.w = None if weights is None else np.asarray(weights, dtype=np.float64)
        self.eigen = eigen                                         # callable: cov6 -> (info, axis) (the oracle's dsyev restatement)
        # distinct (colour, weight) rows: exact sums run over the distinct rows with multiplicities
        rows = self.c.T if self.w is None else np.column_stack([self.c.T, self.w])
        uq, inv = np.unique(rows, axis=0, return_inverse=True)
        self.inv = inv.reshape(-1).astype(np.int64)
        self.U = uq.shape[0]
        self.X = []
        S = 0
        cols = [_exact_ints(uq[:, j]) for j in range(3)]
        S = max(s for _, s in cols)
        self.X = [np.array([int(v) << (S - s) for v in x], dtype=object) for x, s in cols]
        self.S = S
        if self.w is None:
            self.Wt = np.array([1] * self.U, dtype=object)
            self.T = 0
            self.Wfloor = np.array([1] * self.U, dtype=object)
        else:
            self.Wt, self.T = _exact_ints(uq[:, 3])
            self.Wfloor = np.array([int(np.floor(v)) for v in uq[:, 3]], dtype=object)   # size_t += double truncates (local.c:133)
        # per distinct row: w*c_j (scale S+T) and w*|c|^2 (scale 2S+T)
        self.WX = [self.Wt * self.X[j] for j in range(3)]
        self.WX2 = self.Wt * (self.X[0] * self.X[0] + self.X[1] * self.X[1] + self.X[2] * self.X[2])
        self.X2 = self.X[0] * self.X[0] + self.X[1] * self.X[1] + self.X[2] * self.X[2]

    # ---- exact sums over a member set -------------------------------------------------------
    def _counts(self, idx):
        cnt = np.bincount(self.inv[idx], minlength=self.U)
        nz = np.flatnonzero(cnt)
        return nz, np.array([int(v) for v in cnt[nz]], dtype=object)

    def _sums(self, idx, weighted=True):
        nz, cn = self._counts(idx)
        if weighted:
            SW = int((cn * self.Wt[nz]).sum()) if len(nz) else 0
            S1 = [int((cn * self.WX[j][nz]).sum()) if len(nz) else 0 for j in range(3)]
            S2 = int((cn * self.WX2[nz]).sum()) if len(nz) else 0
            SZ = int((cn * self.Wfloor[nz]).sum()) if len(nz) else 0
        else:
            SW = int(cn.sum()) if len(nz) else 0
            S1 = [int((cn * self.X[j][nz]).sum()) if len(nz) else 0 for j in range(3)]
            S2 = int((cn * self.X2[nz]).sum()) if len(nz) else 0
            SZ = SW
        return SW, S1, S2, SZ

    def _dist(self, SW, S1, S2, T):
        """sum w |c - mean|^2 exactly (Fraction), from sums at weight scale T."""
        if SW == 0:
            return Fraction(0)
        q = S1[0] * S1[0] + S1[1] * S1[1] + S1[2] * S1[2]
        return Fraction(S2 * SW - q, SW << (2 * self.S + T))

    def _cov6(self, idx, weighted):
        """exact covariance (xx, yx, zx, yy, zy, zz) about the exact mean, as floats, and its trace."""
        nz, cn = self._counts(idx)
        Wt = self.Wt if weighted else np.array([1] * self.U, dtype=object)
        T = self.T if weighted else 0
        w = cn * Wt[nz]
        SW = int(w.sum())
        S1 = [int((w * self.X[j][nz]).sum()) for j in range(3)]
        out = []
        for (a, b) in ((0, 0), (1, 0), (2, 0), (1, 1), (2, 1), (2, 2)):
            Sab = int((w * self.X[a][nz] * self.X[b][nz]).sum())
            out.append(Fraction(Sab * SW - S1[a] * S1[b], (SW * SW) << (2 * self.S)))
        return [float(v) for v in out], float(out[0] + out[3] + out[5])

    # ---- the reference's f64 bucket rule (sort.c:12-91) --------------------------------------
    def buckets(self, idx, axis):
        c = self.c
        dots = (c[0][idx] * axis[0] + c[1][idx] * axis[1]) + c[2][idx] * axis[2]
        mn, mx = dots.min(), dots.max()
        if mx - mn < DELTA:
            return (np.arange(len(idx)) % BUCKETS).astype(np.int64), True, np.zeros(len(idx), dtype=bool), np.zeros(len(idx), dtype=np.int64)
        s = 1 / (mx - mn)
        pos = BUCKETS * ((dots - mn) * s)
        b = np.minimum(BUCKETS - 1, pos.astype(np.int64))
        # a pixel whose position is within the rounding of the dot product (three products, two sums, the subtraction and the scale)
        # of a bucket border may fall on either side in another evaluation order of the projection (dgemv, sort.c:43-56)
        tol = BUCKETS * 16 * EPS * float(np.abs(dots).max() + abs(mn)) * s + 1e-12
        nearest = np.rint(pos)
        amb = (np.abs(pos - nearest) <= tol) & (nearest >= 1) & (nearest <= BUCKETS - 1)
        other = np.where(nearest > b, b + 1, b - 1)                 # the bucket on the other side of the near border
        other = np.clip(other, 0, BUCKETS - 1)
        return b, False, amb, other

    # ---- cut objective (local.c:136-176) exactly, for every bucket -----------------------------
    def _objectives(self, idx, bkt):
        """-> list over occupied buckets (ascending): (bucket, Fraction objective of cutting AT that bucket)."""
        key = bkt * self.U + self.inv[idx]
        uk, cnt = np.unique(key, return_counts=True)
        per = {}
        for k, c in zip(uk.tolist(), cnt.tolist()):
            b, u = divmod(k, self.U)
            p = per.get(b)
            if p is None:
                p = per[b] = [0, 0, 0, 0]
            p[0] += c * self.WX[0][u]; p[1] += c * self.WX[1][u]; p[2] += c * self.WX[2][u]; p[3] += c * self.Wfloor[u]
        occ = sorted(per)
        tot = [sum(per[b][q] for b in occ) for q in range(4)]
        out = []
        run = [0, 0, 0, 0]
        sc = 1 << (2 * (self.S + self.T))
        for b in occ:
            for q in range(4):
                run[q] += per[b][q]
            sl, sr = run[3], tot[3] - run[3]
            v = Fraction(0)
            for j in range(3):
                csl, csr = run[j], tot[j] - run[j]
                if sl != 0:
                    v += Fraction(csl * csl, sl)
                if sr != 0:
                    v += Fraction(csr * csr, sr)
            out.append((b, v / sc))
        return out

    # ---- one cluster's record ------------------------------------------------------------------
    def _make(self, idx):
        cl = _Cluster()
        cl.idx = idx
        cl.n = len(idx)
        cl.SW, cl.S1, cl.S2, cl.SZ = self._sums(idx)
        cl.dist = self._dist(cl.SW, cl.S1, cl.S2, self.T)
        cl.split_rec = None
        cl.children = None
        cl.benefit = None
        return cl

    def _benefit_of(self, cl, left, right):
        return cl.dist - left.dist - right.dist

    def _own_split(self, cl):
        """The referee's own split_cluster (local.c:179-254) of a cluster no trace record describes: axis by the eigen restatement from
        the exact covariance rounded to f64, buckets by the f64 rule, cut at the EXACT arg-max of the objective."""
        if cl.n <= 1:
            return None
        cov6, _ = self._cov6(cl.idx, True)
        info, axis = self.eigen(cov6)
        if info != 0:
            return None
        b, deg, _, _ = self.buckets(cl.idx, axis)
        objs = self._objectives(cl.idx, b)
        best = max(objs, key=lambda t: t[1])
        left = cl.idx[b <= best[0]]
        right = cl.idx[b > best[0]]
        return self._make(left), self._make(right)

    # ---- global quantiser --------------------------------------------------------------------
    def _check_gq(self, tr, K, rep):
        n = self.n
        all_idx = np.arange(n)
        cov6, trc = self._cov6(all_idx, False)
        _, _, S2all, _ = self._sums(all_idx, False)
        # first order: n rounded addends of size ~trace; second order: the reference centres on a ROUNDED mean (pca.c:33-60), off
        # by ~n eps |c|, whose square enters every entry -- all that is left when the pixels are one colour
        env = ENV * n * EPS * max(trc, 0.0) + ENV * (n * EPS) ** 2 * float(Fraction(S2all, n << (2 * self.S))) + 1e-300
        for q in range(6):
            if abs(cov6[q] - tr["gq_cov6"][q]) > env:
                rep.violations.append("GQ covariance[%d] %.17g vs exact %.17g: off by %.3g > envelope %.3g" %
                                      (q, tr["gq_cov6"][q], cov6[q], abs(cov6[q] - tr["gq_cov6"][q]), env))
        info, ax = self.eigen(tr["gq_cov6"])
        rep.decisions += 1
        if info != 0 or list(ax) != list(tr["gq_axis"]):
            rep.violations.append("GQ axis %r is not dsyev(traced covariance) = %r" % (tr["gq_axis"], list(ax)))
        b, deg, amb, _ = self.buckets(all_idx, tr["gq_axis"])
        rep.ambiguous_members += int(amb.sum())
        cuts = list(tr["gq_cuts"])
        k = tr["n_base"]
        if len(cuts) != k + 1 or cuts[0] != 0 or cuts[-1] != BUCKETS or any(cuts[i] >= cuts[i + 1] for i in range(k)):
            rep.violations.append("GQ cuts malformed: %r" % (cuts,))
            return None
        # cell moments (cells.c:53-139), exact prefix sums
        nzb = np.bincount(b, minlength=BUCKETS)
        w0 = [0] * (BUCKETS + 1)
        w1 = [[0] * (BUCKETS + 1) for _ in range(3)]
        w2 = [0] * (BUCKETS + 1)
        key = b * self.U + self.inv
        uk, cnt = np.unique(key, return_counts=True)
        for kk, c in zip(uk.tolist(), cnt.tolist()):
            bb, u = divmod(kk, self.U)
            w0[bb + 1] += c
            for j in range(3):
                w1[j][bb + 1] += c * self.X[j][u]
            w2[bb + 1] += c * self.X2[u]
        for i in range(1, BUCKETS + 1):
            w0[i] += w0[i - 1]; w2[i] += w2[i - 1]
            for j in range(3):
                w1[j][i] += w1[j][i - 1]
        sc = 1 << (2 * self.S)

        def D(a, bb):                                              # cells.c:141-182, exactly
            cn = w0[bb] - w0[a]
            if cn == 0:
                return Fraction(0)
            q = sum((w1[j][bb] - w1[j][a]) ** 2 for j in range(3))
            return Fraction((w2[bb] - w2[a]) * cn - q, cn * sc)

        def cost(q):                                               # what the DP of global.c:262-288 minimises: a LAST bucket alone counts 0
            tot = D(q[0], q[1])
            for j in range(1, len(q) - 1):
                tot += Fraction(0) if q[j + 1] - q[j] == 1 else D(q[j], q[j + 1])
            return tot

        # the reference's DP in f64 on the exact cell distortions rounded once (global.c:230-288): optimum per k
        Df = np.zeros((BUCKETS + 1, BUCKETS + 1))
        w0f = np.array(w0, dtype=np.float64)
        w2f = np.array([float(Fraction(v, sc)) for v in w2])
        w1f = np.array([[float(Fraction(v, 1 << self.S)) for v in w1[j]] for j in range(3)])
        for a in range(BUCKETS):
            cn = w0f[a + 1:] - w0f[a]
            q = sum((w1f[j][a + 1:] - w1f[j][a]) ** 2 for j in range(3))
            with np.errstate(divide="ignore", invalid="ignore"):
                d = (w2f[a + 1:] - w2f[a]) - np.where(cn > 0, q / np.where(cn > 0, cn, 1), 0.0)
            Df[a, a + 1:] = np.where(cn > 0, d, 0.0)
        S2tot = float(Fraction(w2[BUCKETS], sc))
        env_gq = ENV * n * EPS * max(S2tot, 1e-300)                # the prefix differences cancel against sums of this size
        E = Df[0].copy()
        kmax = min(K, GQ_MAX_K)
        Ek = {1: E.copy()}
        for kk in range(2, kmax + 1):
            Eprev = E.copy()
            for nn in range(kk + 1, BUCKETS + 1):
                cand = Eprev[kk - 1:nn - 1] + Df[kk - 1:nn - 1, nn]
                E[nn] = min(Eprev[nn - 1], cand.min()) if len(cand) else Eprev[nn - 1]
            Ek[kk] = E.copy()
        rep.decisions += 1
        if k > kmax:
            rep.violations.append("GQ made %d base clusters, more than min(K, 12) = %d" % (k, kmax))
        elif k >= 2:
            cexact = cost(cuts)
            gap = float(cexact) - float(Ek[k][BUCKETS])
            if gap > env_gq:
                rep.violations.append("GQ cuts %r cost %.17g, the DP's optimum for k=%d is %.17g: gap %.3g > envelope %.3g" %
                                      (cuts, float(cexact), k, float(Ek[k][BUCKETS]), gap, env_gq))
            elif gap > 0:
                rep.tie("gq_cuts", -1, gap, env_gq)
        # termination (global.c:99-187, tested on the quantiser of k - 1 cells before k is built): the traced quantiser must satisfy
        # the stop rule iff it stopped short of kmax
        if k < kmax:
            stop, why = self._gq_terminates(cuts, tr["gq_axis"], D, w0, w1, w2, b)
            rep.decisions += 1
            if not stop:
                rep.violations.append("GQ stopped at %d < %d base clusters but its quantiser does not meet the stop rule (%s)" % (k, kmax, why))
        lut = np.zeros(BUCKETS, dtype=np.int64)
        for bb in range(BUCKETS):                                   # bucket b belongs to the first cell j with b+1 <= q[j+1] (global.c:328-335)
            j = 0
            while j < k - 1 and not (bb + 1 <= cuts[j + 1]):
                j += 1
            lut[bb] = j
        cell = lut[b]
        return [all_idx[cell == j] for j in range(k)]

    def _gq_terminates(self, cuts, gaxis, D, w0, w1, w2, b):
        """should_terminate (global.c:99-187) on the traced cuts, in f64 from exact cell sums; a bias within 1e-9 of a threshold counts
        either way."""
        dist = [float(D(cuts[j], cuts[j + 1])) for j in range(len(cuts) - 1)]
        total = sum(dist)
        if total < DELTA * (1 + 1e-6):
            return True, "distortion %.3g" % total
        # cell covariance from raw moments (cells.c:184-223) needs the cross moments: recompute from the members
        bias = 0.0
        near = False
        for j in range(len(cuts) - 1):
            members = np.flatnonzero((b >= cuts[j]) & (b < cuts[j + 1]))
            if len(members) == 0:
                cov6 = [0.0] * 6
            else:
                cov6, _ = self._cov6(members, False)
            info, ca = self.eigen(cov6)
            if info != 0:
                return True, "eigen-solver failed"
            norms = float(np.sqrt(sum(v * v for v in gaxis)) * np.sqrt(sum(v * v for v in ca)))
            cb = 0.0 if norms < DELTA else min(1.0, abs(sum(ca[i] * gaxis[i] for i in range(3)) / norms))
            if abs(cb - 0.9) < 1e-9:
                near = True
            if cb < 0.9:
                continue
            bias += dist[j] / total * cb
        if bias < 0.1 or abs(bias - 0.1) < 1e-9 or near:
            return True, "bias %.6g" % bias
        return False, "bias %.6g, distortion %.3g" % (bias, total)

    # ---- the whole trace -----------------------------------------------------------------------
    def check(self, tr, K):
        rep = Report()
        base = self._check_gq(tr, K, rep)
        if base is None:
            return rep
        result = [self._make(ix) for ix in base]
        count = len(result)
        if tr["n_base"] != count:
            rep.violations.append("n_base %d" % tr["n_base"])
            return rep
        # replay the commits: membership of every cluster the trace splits (its partition follows from the traced axis + bucket)
        steps = tr["splits"]
        lookup = {}                                                # membership signature -> the cluster record with its traced children
        initial = list(result)
        for si, s in enumerate(steps):
            rep.decisions += 1
            row = s["row"]
            if not (0 <= row < count) or s["new_row"] != count or count >= K:
                rep.violations.append("step %d: row %d / new_row %d with %d clusters (K=%d)" % (si, row, s["new_row"], count, K))
                return rep
            cl = result[row]
            nn = cl.n
            if nn != s["n"]:
                rep.violations.append("step %d: the cluster has %d members, the trace says %d" % (si, nn, s["n"]))
                return rep
            swf = float(Fraction(cl.SW, 1 << self.T))
            if abs(swf - s["sw"]) > ENV * nn * EPS * swf:
                rep.violations.append("step %d: sum of weights %.17g vs exact %.17g" % (si, s["sw"], swf))
            # covariance and axis
            cov6, trc = self._cov6(cl.idx, True)
            envc = (ENV * nn * EPS * max(trc, 0.0) + ENV * (nn * EPS) ** 2 * float(Fraction(cl.S2, cl.SW << (2 * self.S))) + 1e-300) if cl.SW else 0.0
            for q in range(6):
                if abs(cov6[q] - s["cov6"][q]) > envc:
                    rep.violations.append("step %d: covariance[%d] %.17g vs exact %.17g: off by %.3g > envelope %.3g" %
                                          (si, q, s["cov6"][q], cov6[q], abs(cov6[q] - s["cov6"][q]), envc))
                    break
            info, ax = self.eigen(s["cov6"])
            if info != 0 or list(ax) != list(s["axis"]):
                rep.violations.append("step %d: axis %r is not dsyev(traced covariance) = %r" % (si, s["axis"], list(ax)))
            # members
            b, deg, amb, other = self.buckets(cl.idx, s["axis"])
            if bool(s["degenerate"]) != deg:
                rep.violations.append("step %d: degenerate flag %d, the projection's range says %d" % (si, s["degenerate"], int(deg)))
                return rep
            split = s["split"]
            goes_left = b <= split
            nl = int(goes_left.sum())
            if nl != s["n_left"]:
                # pixels on a bucket border next to the cut may sit on the other side in the run's own evaluation order
                movable = amb & ((other <= split) != goes_left)
                need = int(s["n_left"]) - nl
                cand = np.flatnonzero(movable & (goes_left if need < 0 else ~goes_left))
                if abs(need) != len(cand):
                    rep.violations.append("step %d: cut at bucket %d gives %d | %d members, the trace says %d | %d (%d border pixels could move)" %
                                          (si, split, nl, nn - nl, s["n_left"], s["n_right"], len(cand)))
                    return rep
                goes_left = goes_left.copy()
                goes_left[cand] = need > 0
                b = b.copy()
                b[cand] = other[cand]
                rep.ambiguous_members += len(cand)
            if int(s["n_left"]) + int(s["n_right"]) != nn or s["n_left"] == 0 or s["n_right"] == 0:
                if not (s["n_left"] == 0 or s["n_right"] == 0):
                    rep.violations.append("step %d: member counts %d + %d != %d" % (si, s["n_left"], s["n_right"], nn))
                    return rep
            left, right = self._make(cl.idx[goes_left]), self._make(cl.idx[~goes_left])
            cl.children = (left, right)
            cl.benefit = self._benefit_of(cl, left, right)
            lookup[self._sig(cl)] = cl
            # cut admissibility
            objs = self._objectives(cl.idx, b)
            occ = [t[0] for t in objs]
            chosen = None
            for bb, v in objs:                                      # the objective of cutting at `split` = that of the last occupied bucket <= split
                if bb <= split:
                    chosen = v
            if chosen is None:
                chosen = Fraction(0)                                # nothing goes left: objective of the empty prefix (sl = 0) = that of everything right
                chosen = objs[-1][1]
            vmax = max(v for _, v in objs)
            gap = vmax - chosen
            envo = ENV * nn * EPS * float(vmax)
            if gap > 0:
                if float(gap) > envo:
                    rep.violations.append("step %d: cut at bucket %d has objective %.17g, the maximum is %.17g: gap %.3g > envelope %.3g" %
                                          (si, split, float(chosen), float(vmax), float(gap), envo))
                else:
                    rep.tie("cut", si, gap, envo)
            # traced distortions / benefit
            envd = ENV * nn * EPS * float(cl.dist) + 1e-300
            for name, val in (("dist", cl.dist), ("dist_left", left.dist), ("dist_right", right.dist), ("benefit", cl.benefit)):
                if abs(float(val) - s[name]) > envd:
                    rep.violations.append("step %d: traced %s %.17g vs exact %.17g: off by %.3g > envelope %.3g" %
                                          (si, name, s[name], float(val), abs(float(val) - s[name]), envd))
                    break
            # commit (local.c:375-376)
            result.append(left)
            result[row] = right
            count += 1
        # greedy admissibility: replay once more with every frontier cluster's benefit known
        self._check_greedy(tr, K, initial, lookup, rep)
        return rep

    @staticmethod
    def _sig(cl):
        return (cl.n, int(cl.idx[0]) if cl.n else -1, int(cl.idx[-1]) if cl.n else -1, cl.S1[0], cl.S1[1], cl.S2)

    def _split_of(self, cl, lookup):
        """children + exact benefit of a frontier cluster: from the trace where it is split later, else the referee's own evaluation."""
        if cl.benefit is not None:
            return
        known = lookup.get(self._sig(cl))
        if known is not None:
            cl.children = known.children
            cl.benefit = known.benefit
            return
        ch = self._own_split(cl)
        if ch is None:
            cl.children = None
            cl.benefit = Fraction(0)
        else:
            cl.children = ch
            cl.benefit = self._benefit_of(cl, ch[0], ch[1])

    def _check_greedy(self, tr, K, result, lookup, rep):
        frontier = list(result)
        for cl in frontier:
            self._split_of(cl, lookup)
        for si, s in enumerate(tr["splits"]):
            rep.decisions += 1
            row = s["row"]
            chosen = frontier[row]
            bmax_j, bmax = max(enumerate(c.benefit for c in frontier), key=lambda t: t[1])
            gap = bmax - chosen.benefit
            env = ENV * EPS * (chosen.n * float(chosen.dist) + frontier[bmax_j].n * float(frontier[bmax_j].dist)) + 1e-300
            if gap > 0:
                if float(gap) > env:
                    rep.violations.append("step %d: row %d has benefit %.17g, row %d has %.17g: gap %.3g > envelope %.3g" %
                                          (si, row, float(chosen.benefit), bmax_j, float(bmax), float(gap), env))
                else:
                    rep.tie("greedy", si, gap, env)
            if float(chosen.benefit) < DELTA - env:
                rep.violations.append("step %d: committed a split of benefit %.3g < 1e-16" % (si, float(chosen.benefit)))
            left, right = chosen.children
            self._split_of(left, lookup)
            self._split_of(right, lookup)
            frontier.append(left)
            frontier[row] = right
        count = len(frontier)
        if count != tr["n_clusters"]:
            rep.violations.append("the trace ends with %d clusters, its header says %d" % (count, tr["n_clusters"]))
        if count < K:                                              # stopped early: nothing left may be worth a split (local.c:365-370)
            rep.decisions += 1
            bmax_j, bmax = max(enumerate(c.benefit for c in frontier), key=lambda t: t[1])
            env = ENV * EPS * frontier[bmax_j].n * float(frontier[bmax_j].dist) + 1e-300
            if float(bmax) >= DELTA + env:
                rep.violations.append("stopped at %d < %d clusters although row %d still has benefit %.3g >= 1e-16" %
                                      (count, K, bmax_j, float(bmax)))
            elif float(bmax) >= DELTA:
                rep.tie("stop", len(tr["splits"]), float(bmax) - DELTA, env)


def first_divergence(ta, tb):
    """The first decision two traces disagree on: None, or (kind, step, description)."""
    if ta["n_base"] != tb["n_base"]:
        return ("gq_base_count", -1, "%d vs %d base clusters" % (ta["n_base"], tb["n_base"]))
    sa, sb = np.array(ta["gq_axis"]), np.array(tb["gq_axis"])
    if np.allclose(sa, -sb, atol=1e-9) and not np.allclose(sa, sb, atol=1e-9):
        return ("gq_axis_sign", -1, "global axis %r vs %r" % (ta["gq_axis"], tb["gq_axis"]))
    if ta["gq_cuts"] != tb["gq_cuts"]:
        return ("gq_cuts", -1, "%r vs %r" % (ta["gq_cuts"], tb["gq_cuts"]))
    for i, (a, b) in enumerate(zip(ta["splits"], tb["splits"])):
        if a["row"] != b["row"]:
            return ("greedy", i, "row %d vs %d (benefits %.17g vs %.17g)" % (a["row"], b["row"], a["benefit"], b["benefit"]))
        xa, xb = np.array(a["axis"]), np.array(b["axis"])
        if np.allclose(xa, -xb, atol=1e-9) and not np.allclose(xa, xb, atol=1e-9):
            return ("axis_sign", i, "axis %r vs %r" % (a["axis"], b["axis"]))
        if not np.allclose(xa, xb, atol=1e-9):
            return ("axis", i, "axis %r vs %r" % (a["axis"], b["axis"]))
        if (a["n_left"], a["n_right"]) != (b["n_left"], b["n_right"]):
            return ("cut", i, "bucket %d (%d | %d) vs bucket %d (%d | %d)" % (a["split"], a["n_left"], a["n_right"], b["split"], b["n_left"], b["n_right"]))
    if len(ta["splits"]) != len(tb["splits"]):
        return ("stop", min(len(ta["splits"]), len(tb["splits"])), "%d vs %d commits" % (len(ta["splits"]), len(tb["splits"])))
    return None


def oracle_eigen(ob):
    """cov6 -> (info, principal axis) through the oracle's dsyev restatement (orc_eigen_sym3, pinned against LAPACK)."""
    def f(c6):
        a = np.array([[c6[0], c6[1], c6[2]], [c6[1], c6[3], c6[4]], [c6[2], c6[4], c6[5]]])
        info, w, z = ob.eigen_sym3(a)
        return info, [float(z[0, 2]), float(z[1, 2]), float(z[2, 2])]
    return f


# ---- what the -m gpu tests call when the HIP path's result differs from the oracle's ---------------------------------------
_CONVERT = {1: ("srgb_to_cieluv", 1), 2: ("srgb_to_ictcp", 0)}         # colour space -> (oracle name, PAMD_* code of patolette_amd_convert)


def converted(ob, flat_srgb, cs, gpu=None):
    """The image in the quantisation space (patolette.c:201-207): by the oracle, or -- gpu given -- by the HIP path's own kernel."""
    if cs == 0:
        return np.array(flat_srgb, dtype=np.float64, copy=True)
    if gpu is None:
        return ob.convert(_CONVERT[cs][0], flat_srgb)
    import ctypes as C
    out = np.array(flat_srgb, dtype=np.float64, copy=True)
    rc = gpu.patolette_amd_convert(_CONVERT[cs][1], out.ctypes.data_as(C.POINTER(C.c_double)), out.size // 3)
    assert rc == 0
    return out


def explain_divergence(ob, native, gpu, width, height, flat, wts, K, cs, dither, niter, max_samples, pal_g, map_g):
    """The HIP path's full-path result (pal_g, map_g: the call must be the LAST one on this thread; flat: the planar sRGB image it
    was given) differs from the oracle's.
    Returns a dict describing why that is tie noise -- or raises AssertionError naming the decision that is not.

    1. every decision in the HIP path's split trace is the exact optimum or within the rounding envelope of it (Referee.check on the
       pixels as the HIP path's own conversion kernel produced them);
    2. so is every decision of the oracle's trace (the reference's own choice is inside the envelope);
    3. everything behind the quantisers is deterministic: the oracle's KMeans + mapping / dithering + write-out (patolette.c:246-336)
       started from the HIP path's cluster centres must reproduce the HIP path's palette (1e-9) and map (bit for bit)."""
    n = width * height
    trace_g = native.last_split_trace()
    centres_g = native.last_cluster_centers()
    data_g = converted(ob, flat, cs, gpu)
    data_o = converted(ob, flat, cs)
    eig = oracle_eigen(ob)
    rep_g = Referee(data_g, wts, n, eig).check(trace_g, K)
    assert rep_g.ok, "HIP path: decision outside the rounding envelope -- " + rep_g.summary()
    r = ob.quantize_clusters(data_o, wts, n, K, want_membership=False)
    trace_o = ob.last_split_trace()
    rep_o = Referee(data_o, wts, n, eig).check(trace_o, K)
    assert rep_o.ok, "oracle: decision outside the rounding envelope -- " + rep_o.summary()
    fd = first_divergence(trace_o, trace_g)
    assert len(centres_g) == trace_g["n_clusters"]
    pal_r, map_r = ob.patolette_from_centers(width, height, flat, wts, K, centres_g, dither=dither, color_space=cs,
                                             kmeans_niter=niter, kmeans_max_samples=max_samples)
    assert np.allclose(pal_g, pal_r, rtol=0, atol=1e-9), "behind the quantisers: palette differs from the oracle's replay by %.3g" % \
        float(np.max(np.abs(pal_g - pal_r)))
    if max(width, height) > 1 or not dither:                      # the 1x1 Riemersma walk visits nothing (riemersma.c:452-456)
        bad = int(np.sum(map_g != map_r))
        assert bad == 0, "behind the quantisers: %d of %d map entries differ from the oracle's replay" % (bad, n)
    return dict(first=fd, ties_gpu=len(rep_g.ties), ties_oracle=len(rep_o.ties), decisions=rep_g.decisions,
                kinds=sorted(set(t["kind"] for t in rep_g.ties)), n_clusters=(trace_o["n_clusters"], trace_g["n_clusters"]))
