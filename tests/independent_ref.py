"""An INDEPENDENT second reading of the stages no reference build can pin here (GQ, LQ, NN map, Riemersma walk).

Written from the reference's C sources (cited file:line, /root/reference), NOT from oracle/patolette_oracle.c: numpy
array code with the third-party pieces taken from the real libraries the image has -- `dsyev` and `dgemv` from
scipy's OpenBLAS (what `lib/src/math/eigen.c:83-140` and `lib/src/quantize/sort.c:43-56` call) instead of either of this
repository's restatements.  tests/test_oracle_independent.py runs the oracle against it; a misreading shared by the
oracle and the HIP path's host code would show up there.  Test infrastructure only; small inputs (pure Python loops).

Sequential f64 accumulations of the reference are kept sequential: np.cumsum()[-1] and np.bincount(weights=) add in
index order, np.add.reduce would not.
"""
import numpy as np
from scipy.linalg import blas, lapack

DELTA = 1e-16          # lib/include/math/misc.h:5
BUCKETS = 512          # quantize/global.c:22, local.c:15


def seq_sum(v):
    v = np.asarray(v, dtype=np.float64)
    return float(np.cumsum(v)[-1]) if v.size else 0.0


def eigen_solve(m):
    """math/eigen.c:83-140: dsyev_('V', 'L') on a column-major copy; eigenvalues ascending, eigenvectors as columns."""
    w, v, info = lapack.dsyev(np.asfortranarray(m, dtype=np.float64), compute_v=1, lower=1)
    if info != 0:
        return None, None
    return w, v


def vector_mean(c, w):
    """array/matrix2D.c:200-233: sum of v*w per column, then scaled by 1/rows or 1/sum(w)."""
    if w is None:
        mean = np.array([seq_sum(c[:, j] * 1.0) for j in range(3)])
        return mean * (1 / float(c.shape[0]))
    mean = np.array([seq_sum(c[:, j] * w) for j in range(3)])
    return mean * (1 / seq_sum(w))


def pca_axis(c, w):
    """math/pca.c:33-168: centred copy, vcov[j][k] = sum(weight * cij * cik) / w_sum, dsyev, axis = last column."""
    cen = c - vector_mean(c, w)                                   # pca.c:45-52 (column by column, same subtraction)
    w_sum = float(c.shape[0]) if w is None else seq_sum(w)
    vcov = np.zeros((3, 3))
    ww = np.ones(c.shape[0]) if w is None else w
    for j in range(3):
        for k in range(3):
            vcov[j, k] = seq_sum(ww * cen[:, j] * cen[:, k]) / w_sum      # pca.c:88-95: (weight * cij) * cik
    evals, vecs = eigen_solve(vcov)
    if evals is None:
        return None
    return vecs[:, 2].copy()


def axis_sort(c, axis):
    """quantize/sort.c:12-91: dots by cblas_dgemv (the real one), min/max, round-robin when degenerate, else the bucket formula."""
    dots = blas.dgemv(1.0, np.asfortranarray(c), axis)
    mn, mx = dots.min(), dots.max()
    n = c.shape[0]
    if mx - mn < DELTA:
        return (np.arange(n) % BUCKETS).astype(np.int64)          # sort.c:61-79: j = 0..bucket_count-1 cyclically
    s = 1 / (mx - mn)
    ratio = (dots - mn) * s
    return np.minimum((BUCKETS * ratio).astype(np.int64), BUCKETS - 1)


class Cells:
    """quantize/cells.c:53-139: per-bucket moments, 1-based, made cumulative."""

    def __init__(self, c, bmap):
        size = BUCKETS + 1
        j = bmap + 1
        self.w0 = np.cumsum(np.bincount(j, minlength=size)).astype(np.uint64)
        self.w1 = np.stack([np.cumsum(np.bincount(j, weights=c[:, r], minlength=size)) for r in range(3)])
        self.w2 = np.cumsum(np.bincount(j, weights=(c[:, 0] ** 2 + c[:, 1] ** 2) + c[:, 2] ** 2, minlength=size))
        self.wrs = {}
        for s in range(3):
            for r in range(s + 1):
                self.wrs[(r, s)] = np.cumsum(np.bincount(j, weights=c[:, r] * c[:, s], minlength=size))

    def distortion(self, a, b):                                   # cells.c:141-182
        if self.w0[a] == self.w0[b]:
            return 0.0
        q = self.w1[:, b] - self.w1[:, a]
        return self.w2[b] - self.w2[a] - ((q[0] ** 2 + q[1] ** 2) + q[2] ** 2) / float(self.w0[b] - self.w0[a])

    def distortion_to(self, a, b):
        """distortion(a[i], b) for an array of lower ends: the expression of cells.c:141-182 element-wise."""
        q = self.w1[:, b][:, None] - self.w1[:, a]
        n = (self.w0[b] - self.w0[a]).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            d = self.w2[b] - self.w2[a] - ((q[0] ** 2 + q[1] ** 2) + q[2] ** 2) / n
        return np.where(self.w0[a] == self.w0[b], 0.0, d)

    def vcov(self, a, b):                                         # cells.c:184-259
        m = np.zeros((3, 3))
        if self.w0[a] != self.w0[b]:
            n = float(self.w0[b] - self.w0[a])
            for s in range(3):
                for r in range(s + 1):
                    e = (self.wrs[(r, s)][b] - self.wrs[(r, s)][a]) / n - \
                        (self.w1[r, b] - self.w1[r, a]) * (self.w1[s, b] - self.w1[s, a]) / (n ** 2)
                    m[r, s] = e
        m[2, 0] = m[0, 2]; m[1, 0] = m[0, 1]; m[2, 1] = m[1, 2]   # noqa: E702
        return m

    def axis(self, a, b):                                         # cells.c:261-278
        evals, vecs = eigen_solve(self.vcov(a, b))
        return None if evals is None else vecs[:, 2].copy()

    def bias(self, a, b, axis):                                   # cells.c:280-328
        ca = self.axis(a, b)
        if ca is None:
            return -1.0
        norm = lambda v: np.sqrt((v[0] ** 2 + v[1] ** 2) + v[2] ** 2)       # noqa: E731
        norms = norm(axis) * norm(ca)
        if norms < DELTA:
            return 0.0
        dot = ca[0] * axis[0] + ca[1] * axis[1] + ca[2] * axis[2]
        return min(1.0, abs(dot / norms))


def should_terminate(q, axis, cells):
    """quantize/global.c:99-187"""
    dist = 0.0
    for j in range(len(q) - 1):
        dist += cells.distortion(q[j], q[j + 1])
    if dist < DELTA:
        return True
    bias = 0.0
    for i in range(len(q) - 1):
        cd = cells.distortion(q[i], q[i + 1])
        cb = cells.bias(q[i], q[i + 1], axis)
        if cb < 0:
            return True
        if cb < 0.9:                                              # cell_bias_threshold, global.c:21
            continue
        bias += (cd / dist) * cb
    return bias < 0.1                                             # bias_threshold, global.c:20


def principal_quantizer(K, cells):
    """quantize/global.c:189-298: DP over the 512 buckets, k = 2..min(12, K), with the termination test BEFORE each k."""
    N = BUCKETS
    axis = cells.axis(0, N)
    if axis is None:
        return None
    E = np.zeros(N + 1)
    for i in range(1, N + 1):
        E[i] = cells.distortion(0, i)
    L = {(i, i): i for i in range(1, K + 1)}                     # global.c:236-238

    def chain(k):                                                 # l_chain, global.c:72-97
        ch = [0] * (k + 1)
        t = N
        for j in range(k - 1, 0, -1):
            t = L[(j + 1, t)]
            ch[j] = t
        ch[k] = N
        return ch
    result = chain(1)
    for k in range(2, min(12, K) + 1):
        if should_terminate(result, axis, cells):
            break
        E_ = E.copy()
        for n in range(k + 1, N + 1):
            # global.c:252-272: cut = n-1, e = E_[n-1]; then t = n-2 down to k-1: `if (c < e)` takes it.  A strict '<' scanned
            # downwards keeps the FIRST minimum of (E_[n-1], c(n-2), c(n-3), ...), which is what argmin returns; the candidates
            # are evaluated element-wise with the same expression as Cells.distortion (no reassociation).
            ts = np.arange(n - 2, k - 2, -1)
            cand = np.concatenate(([E_[n - 1]], E_[ts] + cells.distortion_to(ts, n)))
            j = int(np.argmin(cand))
            L[(k, n)] = n - 1 if j == 0 else int(ts[j - 1])
            E[n] = cand[j]
        result = chain(k)
    return result


def gq_quantize(c, w, K):
    """quantize/global.c:388-443 + :300-377: clusters as ascending index lists, in quantizer order."""
    axis = pca_axis(c, None)                                      # the global PCA is UNWEIGHTED (global.c:407)
    if axis is None:
        return None
    bmap = axis_sort(c, axis)
    q = principal_quantizer(K, Cells(c, bmap))
    if q is None:
        return None
    count = len(q) - 1
    owner = np.zeros(BUCKETS, dtype=np.int64)
    for b in range(BUCKETS):
        for j in range(count):
            if b + 1 <= q[j + 1]:
                owner[b] = j
                break
    cl = owner[bmap]
    return [np.nonzero(cl == j)[0] for j in range(count)]


def cluster_center(c, w, idx):
    return vector_mean(c[idx], None if w is None else w[idx])      # cluster.c:171-189


def cluster_distortion(c, w, idx):
    """quantize/cluster.c:111-152"""
    if len(idx) == 0:
        return 0.0
    x = cluster_center(c, w, idx)
    cc = c[idx]
    d = ((cc[:, 0] - x[0]) ** 2 + (cc[:, 1] - x[1]) ** 2) + (cc[:, 2] - x[2]) ** 2
    return seq_sum(d if w is None else d * w[idx])


def split_cluster(c, w, idx, extended=False):
    """quantize/local.c:102-254: weighted axis, 512 buckets, objective over every cut, first maximum.
    extended=True evaluates the objective in long double from exactly-summed prefixes instead (an arg-max over all 512 cuts
    that does not share the reference's f64 rounding): used to check that the decision is not a rounding artefact."""
    if len(idx) <= 1:
        return None
    cc = c[idx]
    ww = None if w is None else w[idx]
    axis = pca_axis(cc, ww)
    if axis is None:
        return None
    bmap = axis_sort(cc, axis)
    wt = np.ones(len(idx)) if ww is None else ww
    ft = np.longdouble if extended else np.float64
    sums = np.stack([np.cumsum(np.bincount(bmap, weights=cc[:, j] * wt, minlength=BUCKETS).astype(ft)) for j in range(3)])
    # local.c:133: `sizes[bucket] += weight` on a size_t: the sum is converted back to an integer at every step
    sizes = np.zeros(BUCKETS, dtype=np.uint64)
    if ww is None:
        sizes = np.bincount(bmap, minlength=BUCKETS).astype(np.uint64)
    else:
        order = np.argsort(bmap, kind="stable")                  # per bucket, members in index order
        bs, xs = bmap[order], ww[order]
        starts = np.flatnonzero(np.r_[True, bs[1:] != bs[:-1]])
        for a, e in zip(starts, np.r_[starts[1:], len(bs)]):
            acc = 0
            for x in xs[a:e].tolist():
                acc = int(float(acc) + x)                         # size_t += double: converted back to an integer every step
            sizes[bs[a]] = np.uint64(acc)
    sizes = np.cumsum(sizes)
    # local.c:147-170, all 512 cuts at once (element-wise: the same operations in the same order per cut):
    # objective[i] += (sl != 0 ? csl*csl/sl : 0) + (sr != 0 ? csr*csr/sr : 0), channel after channel
    obj = np.zeros(BUCKETS, dtype=ft)
    sl = sizes.astype(ft)
    sr = ft(sizes[BUCKETS - 1]) - sl
    for j in range(3):
        csl = sums[j]
        csr = sums[j, BUCKETS - 1] - csl
        with np.errstate(divide="ignore", invalid="ignore"):
            v = np.where(sl != 0, (csl * csl) / sl, ft(0)) + np.where(sr != 0, (csr * csr) / sr, ft(0))
        obj = obj + v
    split = int(np.argmax(obj))                                   # vector.c:26-46 maxloc: strict '>' upwards = first maximum
    left = idx[bmap <= split]
    right = idx[bmap > split]
    return left, right


def lq_quantize(c, w, clusters, K):
    """quantize/local.c:318-404: greedy on the split benefit, result[i] = left, result[best] = right."""
    result = list(clusters)
    children = [split_cluster(c, w, cl) for cl in result]
    dist = {}

    def D(ix):
        key = ix.tobytes()
        if key not in dist:
            dist[key] = cluster_distortion(c, w, ix)
        return dist[key]

    def benefit(i):
        ch = children[i]
        if ch is None:
            return 0.0
        return D(result[i]) - (D(ch[0]) + D(ch[1]))
    # local.c:277-307 recomputes every benefit at every step from cached distortions: the values of untouched entries do
    # not change, so they are kept and only the two entries a step rewrites are evaluated again
    ben = [benefit(j) for j in range(len(result))]
    for i in range(len(clusters), K):
        best = int(np.argmax(np.array(ben)))                      # first maximum
        if ben[best] < DELTA:
            break
        left, right = children[best]
        result.append(left)
        children.append(None)
        result[best] = right
        children[i] = split_cluster(c, w, left)
        children[best] = split_cluster(c, w, right)
        ben.append(benefit(i))
        ben[best] = benefit(best)
    return result


def quantize_clusters(c, w, K):
    """GQ + LQ + PALETTE_create (palette/create.c:11-33): centres in palette order and the membership of every colour."""
    base = gq_quantize(c, w, K)
    if base is None:
        return None
    final = lq_quantize(c, w, base, K)
    centers = np.array([cluster_center(c, w, ix) for ix in final])
    member = np.zeros(c.shape[0], dtype=np.int64)
    for j, ix in enumerate(final):
        member[ix] = j
    return centers, member, len(base)


def nn_map(c, pal):
    """palette/nearest.c:150-209 with FLANN's exact search read as: smallest ((d0^2 + d1^2) + d2^2), lowest index on ties."""
    d = ((c[:, None, 0] - pal[None, :, 0]) ** 2 + (c[:, None, 1] - pal[None, :, 1]) ** 2) + (c[:, None, 2] - pal[None, :, 2]) ** 2
    return np.argmin(d, axis=1)


# ---- Riemersma (dither/riemersma.c) -----------------------------------------------------------
NONE, UP, LEFT, RIGHT, DOWN = range(5)


def hilbert_walk(width, height):
    """riemersma.c:124-257 + :437-459: the pixels in visiting order (x, y unsigned: a step off the left / top edge wraps and is skipped)."""
    mx = max(width, height)
    level, value = 0, mx
    while value > 1:
        value >>= 1
        level += 1
    if (1 << level) < mx:
        level += 1
    out = []
    pos = [0, 0]
    M = 1 << 64

    def move(d):
        x, y = pos
        if x < width and y < height:
            out.append((x, y))
        if d == LEFT:
            pos[0] = (x - 1) % M
        elif d == RIGHT:
            pos[0] = (x + 1) % M
        elif d == UP:
            pos[1] = (y - 1) % M
        elif d == DOWN:
            pos[1] = (y + 1) % M

    def trav(lv, d):
        if lv == 1:
            seq = {LEFT: (RIGHT, DOWN, LEFT), RIGHT: (LEFT, UP, RIGHT), UP: (DOWN, RIGHT, UP), DOWN: (UP, LEFT, DOWN)}[d]
            for m in seq:
                move(m)
            return
        plan = {LEFT: (UP, RIGHT, LEFT, DOWN, LEFT, LEFT, DOWN), RIGHT: (DOWN, LEFT, RIGHT, UP, RIGHT, RIGHT, UP),
                UP: (LEFT, DOWN, UP, RIGHT, UP, UP, RIGHT), DOWN: (RIGHT, UP, DOWN, LEFT, DOWN, DOWN, LEFT)}[d]
        trav(lv - 1, plan[0]); move(plan[1]); trav(lv - 1, plan[2]); move(plan[3]); trav(lv - 1, plan[4]); move(plan[5]); trav(lv - 1, plan[6])   # noqa: E702
    if level > 0:
        import sys
        sys.setrecursionlimit(10000)
        trav(level, UP)
        move(NONE)
    return out


def dither(img, width, height, pal):
    """riemersma.c:275-341 + :360-426 + palette/nearest.c:72-148: img (N,3) linear Rec2020 row-scan, pal (k,3)."""
    Rw, Gw, Bw = 0.51254268114958, 0.8234075540095561, 0.2435159132377184
    fw = np.array([np.float64(np.float32(Rw)), np.float64(np.float32(Gw)), np.float64(np.float32(Bw))])   # index built with (float) weights
    palw = pal * fw
    m = np.exp(np.log(16.0) / (16.0 - 1))
    wts = np.zeros(16)
    v = 1.0
    for i in range(16):
        wts[i] = v / 16.0
        v *= m
    q = np.zeros((16, 3))
    out = np.full(width * height, -1, dtype=np.int64)
    for (x, y) in hilbert_walk(width, height):
        err = np.zeros(3)
        for i in range(16):
            err = err + q[i] * wts[i]
        p = img[y * width + x]
        cor = p + err
        qq = np.array([Rw * cor[0], Gw * cor[1], Bw * cor[2]])
        d = ((qq[0] - palw[:, 0]) ** 2 + (qq[1] - palw[:, 1]) ** 2) + (qq[2] - palw[:, 2]) ** 2
        idx = int(np.argmin(d))
        out[y * width + x] = idx
        q[:-1] = q[1:]
        q[15] = p - pal[idx]
    return out


# ---- patolette() (lib/src/patolette.c:157-343): stage sequencing and colour-space routing ------------------------------
def patolette(width, height, colors, weights, K, dither, palette_only, color_space, kmeans_niter, kmeans_max_samples, conv, kmeans):
    """The orchestrator read from patolette.c:157-343, over this file's GQ / LQ / NN map / dither.  `conv(name, (n,3))` and
    `kmeans(colors, weights, centers, niter, max_samples)` are the two stages reference builds DO pin in this image (the
    colour code, `oracle/_ref/libref_color.so`; faiss, `libref_faiss.so`): the caller passes them in.  Returns
    (palette (K,3) with -1 rows for unset entries, map or None)."""
    SRGB, LUV, ICTCP = 0, 1, 2
    c = np.array(colors, dtype=np.float64, copy=True)            # Matrix2D_init copies (patolette.c:187-191)
    if color_space == LUV:                                        # :201-207
        c = conv("srgb_to_cieluv", c)
    elif color_space == ICTCP:
        c = conv("srgb_to_ictcp", c)
    q = quantize_clusters(c, weights, K)                          # :213-245 GQ then LQ
    if q is None:
        return None
    centers, member, _ = q
    pal = kmeans(c, weights, centers, kmeans_niter, kmeans_max_samples) if kmeans_niter > 0 else centers   # :247-264
    pmap = None
    if not palette_only:                                          # :266 -- palette_only skips the map AND the back-conversion
        if dither:                                                # :267-299
            to_rec = {LUV: "cieluv_to_rec2020", ICTCP: "ictcp_to_rec2020", SRGB: "srgb_to_rec2020"}[color_space]
            c = conv(to_rec, c)
            pal = conv(to_rec, pal)
            pmap = dither_map(c, width, height, pal)
            pal = conv("rec2020_to_srgb", pal)
        else:                                                     # :300-324
            if color_space == LUV:                                # "pretty ugly": Luv -> Rec2020 -> sRGB -> ICtCp, pixels and palette
                for name in ("cieluv_to_rec2020", "rec2020_to_srgb", "srgb_to_ictcp"):
                    c = conv(name, c)
                    pal = conv(name, pal)
            pmap = nn_map(c, pal)
            # :322-323 unconditionally: in the sRGB colour space the palette (sRGB values) goes through the ICtCp -> Rec2020 ->
            # sRGB conversions all the same
            pal = conv("ictcp_to_rec2020", pal)
            pal = conv("rec2020_to_srgb", pal)
    out = np.full((K, 3), -1.0)                                   # :327-336
    out[:len(pal)] = pal
    return out, pmap


def dither_map(img, width, height, pal):
    out = dither(img, width, height, pal)
    if max(width, height) <= 1:
        return None                                               # riemersma.c:452-456: a 1x1 image is never visited
    return out
