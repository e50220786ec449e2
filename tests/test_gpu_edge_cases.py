"""Edge cases of the drop-in call on the GPU vs the CPU oracle: degenerate images, tiny sizes,
K larger than the number of pixels / colours, every colour-space x mapping branch,
palette_only, extreme weights."""
import ctypes as C

import numpy as np
import pytest

from tests.util import match_rows_up_to_permutation

pytestmark = pytest.mark.gpu
dp = C.POINTER(C.c_double)
zp = C.POINTER(C.c_size_t)


def _d(a):
    return a.ctypes.data_as(dp) if a is not None else None


def run_both(native, ob, w, h, flat, wt, K, **kw):
    L = native.lib()
    opts = native.QuantizationOptions(kw.get("dither", False), kw.get("palette_only", False), kw.get("color_space", 2),
                                      kw.get("kmeans_niter", 0), kw.get("kmeans_max_samples", 512 ** 2), False)
    pal = np.zeros((K, 3), dtype=np.float64, order="F")
    n = w * h
    pmap = np.full(n, 12345, dtype=np.uintp)
    code = C.c_int(99)
    L.patolette(w, h, _d(flat), _d(wt), K, C.byref(opts), pal.ctypes.data_as(dp),
                None if kw.get("palette_only", False) else pmap.ctypes.data_as(zp), C.byref(code))
    ec, pal_o, map_o = ob.patolette(w, h, flat, wt, K, dither=kw.get("dither", False), palette_only=kw.get("palette_only", False),
                                    color_space=kw.get("color_space", 2), kmeans_niter=kw.get("kmeans_niter", 0),
                                    kmeans_max_samples=kw.get("kmeans_max_samples", 512 ** 2))
    return (code.value, pal, pmap), (ec, np.asarray(pal_o), map_o)


def assert_same(got, want, ordered=True, tol=1e-9, check_map=True):
    (ec, pal, pmap), (ec_o, pal_o, map_o) = got, want
    assert ec == ec_o == 0
    assert np.array_equal(pal == -1, pal_o == -1)
    if ordered:
        assert np.allclose(pal, pal_o, rtol=0, atol=tol, equal_nan=True), float(np.nanmax(np.abs(pal - pal_o)))
        if check_map and map_o is not None:
            assert np.array_equal(pmap, map_o), int(np.sum(pmap != map_o))
    else:
        perm = match_rows_up_to_permutation(pal, pal_o, tol)
        assert perm is not None
        if check_map and map_o is not None:
            inv = np.empty_like(perm)
            inv[perm] = np.arange(len(perm))
            assert np.array_equal(inv[pmap.astype(np.int64)], map_o.astype(np.int64))


@pytest.mark.parametrize("cs", [0, 1, 2])
@pytest.mark.parametrize("dither", [False, True])
def test_every_colour_space_and_mapping_branch(gpu, native, ob, cs, dither):
    w, h, K = 61, 47, 20
    flat = ob.image(w * h, 30 + cs)
    got, want = run_both(native, ob, w, h, flat, None, K, color_space=cs, dither=dither)
    if cs == 0 and not dither:
        # reference quirk: sRGB + NN pushes an sRGB palette through ICtCp->sRGB; garbage in, same garbage out
        assert_same(got, want, tol=1e-6)
    else:
        assert_same(got, want)


def test_palette_only_returns_quantisation_space_palette(gpu, native, ob):
    w, h, K = 50, 40, 16
    flat = ob.image(w * h, 41)
    for cs in (1, 2):
        got, want = run_both(native, ob, w, h, flat, None, K, color_space=cs, palette_only=True)
        assert got[0] == want[0] == 0
        assert np.allclose(got[1], want[1], rtol=0, atol=1e-9 * max(1.0, float(np.max(np.abs(want[1])))))
        assert np.all(got[2] == 12345)                       # map untouched (NULL passed)


@pytest.mark.parametrize("shape", [(1, 1), (2, 1), (1, 3), (2, 2), (3, 5), (7, 1)])
def test_tiny_images(gpu, native, ob, shape):
    w, h = shape
    flat = ob.image(w * h, 50 + w + 10 * h)
    for K in (1, 2, 4, 64):
        got, want = run_both(native, ob, w, h, flat, None, K)
        # clusters of two pixels are exactly collinear: order ambiguous (tests/util.py)
        assert_same(got, want, ordered=False)


def test_constant_image_and_two_colour_image(gpu, native, ob):
    w, h = 40, 30
    n = w * h
    const = np.concatenate([np.full(n, 0.25), np.full(n, 0.5), np.full(n, 0.75)])
    for K in (1, 5):
        got, want = run_both(native, ob, w, h, const, None, K)
        assert_same(got, want, ordered=False)
        assert int((got[1][:, 0] != -1).sum()) == 1          # a single colour -> a single palette row
    two = const.copy()
    two[:n // 3] = 0.9                                       # red channel differs on a third of the pixels
    got, want = run_both(native, ob, w, h, two, None, 8)
    assert_same(got, want, ordered=False)
    assert int((got[1][:, 0] != -1).sum()) == 2


def test_more_colours_requested_than_pixels(gpu, native, ob):
    w, h, K = 6, 5, 100
    flat = ob.image(w * h, 60)
    got, want = run_both(native, ob, w, h, flat, None, K)
    assert int((want[1][:, 0] != -1).sum()) <= w * h
    assert_same(got, want, ordered=False)


def test_extreme_weights(gpu, native, ob):
    w, h, K = 64, 48, 24
    n = w * h
    flat = ob.image(n, 70)
    wt = 1.0 + 255.0 * ob.weights(n, 70) ** 4 / 256.0        # saliency-like: most ~1, a few up to ~250
    wt[::97] = 1000.0
    for cs, dither in ((1, True), (2, False)):
        got, want = run_both(native, ob, w, h, flat, wt, K, color_space=cs, dither=dither)
        assert_same(got, want)


def test_kmeans_with_fewer_samples_than_min_and_k_not_multiple_of_8(gpu, native, ob):
    w, h, K = 33, 31, 13
    flat = ob.image(w * h, 80)
    got, want = run_both(native, ob, w, h, flat, None, K, kmeans_niter=5, kmeans_max_samples=100)
    assert_same(got, want, tol=1e-6)
    # n == k corner case of faiss (centroids = first k samples): 4x4 image, 16 colours requested and reached
    flat = ob.image(16, 81)
    got, want = run_both(native, ob, 4, 4, flat, None, 16, kmeans_niter=3)
    assert got[0] == want[0] == 0


def test_kmeans_beyond_4096_palette_entries_matches_oracle(gpu, native, ob):
    """The limit round 3 still had (KMeans stopped, loudly, at palette_size 4096) is gone: the reference has none (refine.c:77-89)."""
    w, h, K = 128, 128, 5000
    flat = ob.image(w * h, 90)
    got, want = run_both(native, ob, w, h, flat, None, K, kmeans_niter=2)
    # ~3 pixels per cluster: the last splits divide two-pixel clusters, whose left / right order is the sign dsyev gives a
    # rank-1 covariance (undefined) -- two rows come out swapped; compared up to that permutation
    assert_same(got, want, ordered=False)


@pytest.mark.parametrize("cs,amp", [(2, 1e-3), (2, 1e-4), (2, 3e-5), (1, 1e-4), (1, 3e-5)])
def test_low_variance_image_root_covariance(gpu, native, ob, cs, amp):
    """A nearly flat image: the root covariance as S2 - S1 S1' / n (the moments that ride with the conversion) cancels most of
    its digits here (trace(cov) / trace(S2) ~ 1e-8 .. 1e-11: the pipeline then takes the centred sweep) -- axis, buckets and
    everything after them as the oracle's.  Not below a spread of ~1e-5: there the REFERENCE's own split objective (uncentred
    f64 sums per bucket, csl^2 / sl, local.c:118-176) varies by (spread)^2 ~ 1e-10 relative across the cuts against ~1e-13
    of summation noise, and its arg-max is decided by that noise (measured: 70-549 of 196 608 map entries differ at 1e-5,
    with and without the fallback)."""
    w, h, K = 512, 384, 16
    n = w * h
    flat = np.clip(np.repeat([0.6, 0.35, 0.2], n) + amp * (ob.image(n, 91) - 0.5), 0.0, 1.0)
    got, want = run_both(native, ob, w, h, flat, None, K, color_space=cs)
    assert_same(got, want)


def test_kmeans_skipped_when_a_value_is_not_finite_as_float(gpu, native, ob):
    """faiss scans the f32 training set for NaN / Inf and throws (Clustering.cpp:295-304); refine.c:91 swallows the
    exception, so the palette stays the (float-rounded) initial centres.  A pixel of 1e39 is finite in f64 and Inf once
    cast to float.  (Such an outlier spans more than the ~80 bits of dynamic range the order-independent sums of the HIP
    split loop keep below the largest value -- DESIGN.md section 2 -- so the palettes themselves are not compared with the
    oracle here: the point is that KMeans leaves them alone.)"""
    w, h, K = 64, 48, 12
    n = w * h
    flat = ob.image(n, 3)
    flat[n + 17] = 1e39
    L = native.lib()

    def hip(niter):
        opts = native.QuantizationOptions(False, True, 0, niter, 65536, False)
        pal = np.zeros((K, 3), dtype=np.float64, order="F")
        code = C.c_int(99)
        L.patolette(w, h, _d(flat), None, K, C.byref(opts), pal.ctypes.data_as(dp), None, C.byref(code))
        assert code.value == 0
        return pal
    pal0, pal4 = hip(0), hip(4)
    with np.errstate(over="ignore"):
        assert np.array_equal(pal4, np.where(pal0 == -1, -1.0, pal0.astype(np.float32).astype(np.float64)))
    flat[n + 17] = 0.5                                          # the same image without the outlier: KMeans does move the palette
    assert not np.array_equal(hip(4), hip(0))
    # the stage on its own: centres come back float-rounded and otherwise untouched
    cent = ob.image(K, 4).reshape(3, K).T.copy()
    cio = np.ascontiguousarray(cent.T).reshape(-1).copy()
    flat[5] = np.nan
    assert native.lib().patolette_amd_kmeans_refine(_d(flat), None, n, _d(cio), K, 3, 65536) == 0
    assert np.array_equal(cio.reshape(3, K).T, cent.astype(np.float32).astype(np.float64))


def test_product_eigen_solver_vs_lapack_golden_and_oracle(gpu, native, ob):
    """The split loop's host-side 3x3 eigen-solve (host_math.h, what pipeline.hip calls) against the LAPACK the reference
    calls (eigen.c:83-140 -> dsyev; golden from OpenBLAS 0.3.28) incl. the eigenvector SIGN, and bit for bit against the
    oracle's restatement on 20 000 random covariance-like matrices."""
    from tests.util import bits, golden
    L = native.lib()

    def solve(a):
        buf = np.asfortranarray(np.array(a, dtype=np.float64)).reshape(-1, order="F").copy()
        wv, z = np.zeros(3), np.zeros(9)
        info = L.patolette_amd_eigen_sym3(_d(buf), _d(wv), _d(z))
        return info, wv, z.reshape(3, 3, order="F")
    g = golden("eigen_lapack.npz")
    exact = 0
    for a, wv, v in zip(g["A"], g["W"], g["V"]):
        info, w2, v2 = solve(a)
        assert info == 0
        scale = max(1e-300, np.max(np.abs(wv)))
        assert np.max(np.abs(wv - w2)) <= 1e-13 * scale
        assert np.max(np.abs(v[:, 2] - v2[:, 2])) < 1e-9
        if np.all(np.diff(wv) > 1e-6 * scale):
            assert np.max(np.abs(v - v2)) < 1e-8
        exact += np.array_equal(bits(v), bits(v2))
        ax = np.zeros(3)
        c6 = np.array([a[0, 0], a[1, 0], a[2, 0], a[1, 1], a[2, 1], a[2, 2]])
        assert L.patolette_amd_principal_axis(_d(c6), _d(ax)) == 0 and np.array_equal(bits(ax), bits(v2[:, 2]))
    assert exact > 0.4 * len(g["A"])
    rng = np.random.default_rng(0)
    for _ in range(20000):
        m = rng.normal(size=(3, 3)) * 10.0 ** rng.integers(-6, 3)
        a = m @ m.T
        i1, w1, v1 = solve(a)
        i2, w2, v2 = ob.eigen_sym3(a)
        assert i1 == i2 and np.array_equal(bits(w1), bits(w2)) and np.array_equal(bits(v1), bits(v2))


def test_device_eigen_solver_is_the_hosts_bit_for_bit(gpu, native, ob):
    """The split loop's control kernel solves the children's 3x3 problems ON THE DEVICE (pipeline.hip k_lq_control -> hm::eigen_sym3, the
    host's code compiled for gfx950, one problem per lane).  Same bits as the host / the oracle's dsyev restatement on the LAPACK
    golden matrices, on 20 000 random covariance-like ones, and on rank-deficient / axis-aligned / zero ones (eigen.c:83-140)."""
    import ctypes as C
    from tests.util import bits, golden
    L = native.lib()
    rng = np.random.default_rng(1)
    mats = [np.array(a, dtype=np.float64) for a in golden("eigen_lapack.npz")["A"]]
    for _ in range(20000):
        m = rng.normal(size=(3, 3)) * 10.0 ** rng.integers(-6, 3)
        mats.append(m @ m.T)
    for _ in range(2000):                                        # rank one, rank two, axis-aligned, zero, tiny off-diagonals
        kind = rng.integers(0, 5)
        v = rng.normal(size=3)
        if kind == 0:
            a = np.outer(v, v)
        elif kind == 1:
            u = rng.normal(size=3)
            a = np.outer(v, v) + np.outer(u, u)
        elif kind == 2:
            a = np.diag(np.abs(rng.normal(size=3)))
        elif kind == 3:
            a = np.zeros((3, 3))
        else:
            a = np.diag(np.abs(rng.normal(size=3))) + 1e-17 * (np.outer(v, v))
        mats.append(a * 10.0 ** rng.integers(-12, 2))
    count = len(mats)
    flat = np.concatenate([np.asfortranarray(a).reshape(-1, order="F") for a in mats])
    w = np.zeros(3 * count)
    z = np.zeros(9 * count)
    info = np.zeros(count, dtype=np.int32)
    assert L.patolette_amd_eigen_sym3_device(_d(flat), count, _d(w), _d(z), info.ctypes.data_as(C.POINTER(C.c_int))) == 0
    for i, a in enumerate(mats):
        i2, w2, v2 = ob.eigen_sym3(a)
        assert info[i] == i2
        assert np.array_equal(bits(w[3 * i:3 * i + 3]), bits(w2)), (i, a)
        assert np.array_equal(bits(z[9 * i:9 * i + 9].reshape(3, 3, order="F")), bits(v2)), (i, a)


def test_release_workspace_and_thread_engines(gpu, native, ob):
    """Engines are pooled: a short-lived thread hands its engine back at exit, release_workspace frees the idle ones, and
    the next call works (and gives the same answer) on a fresh workspace."""
    import threading
    import patolette_amd as p
    w, h, K = 80, 60, 10
    colors = ob.image(w * h, 21).reshape(3, -1).T.copy()
    first = p.quantize(w, h, colors, K, dither=False, tile_size=0, kmeans_niter=2)
    box = []
    for _ in range(3):
        t = threading.Thread(target=lambda: box.append(p.quantize(w, h, colors, K, dither=False, tile_size=0, kmeans_niter=2)))
        t.start()
        t.join()
    native.lib().patolette_amd_release_workspace()
    box.append(p.quantize(w, h, colors, K, dither=False, tile_size=0, kmeans_niter=2))
    for r in box:
        assert r[0] and np.array_equal(r[1], first[1]) and np.array_equal(r[2], first[2])


@pytest.mark.parametrize("frac,force_lists", [(0.04, False), (0.08, False), (0.5, False), (0.3, True)])
def test_dominant_colour_long_centroid_chains(gpu, native, ob, monkeypatch, frac, force_lists):
    """A colour that covers a good part of the image means ONE very long centroid chain.  Up to 16 384 samples in one cluster
    (4 % of this image: 10 240) the list-collecting update sums such a list in pieces of 4096 with the block-parallel exact
    form of the chain; beyond that the refinement takes the sorted path (predicted from the size of the largest cluster the
    centres come from).  Either way it must give what the oracle's sequential sums give."""
    if force_lists:                                            # 77 000 members through the list path: 19 pieces, long runs copied block-wide
        monkeypatch.setenv("PAMD_KM_LIST_LONGEST", "100000000")
    w, h, K = 640, 400, 64
    n = w * h
    rng = np.random.default_rng(23)
    rows = rng.random((n, 3))
    k = int(frac * n)
    rows[rng.permutation(n)[:k]] = np.array([0.12, 0.3, 0.65]) + 0.003 * rng.standard_normal((k, 3))
    flat = np.ascontiguousarray(np.clip(rows, 0, 1).T).reshape(-1)
    got, want = run_both(native, ob, w, h, flat, None, K, kmeans_niter=6)
    assert_same(got, want, tol=1e-6)
    st = native.last_stats()
    assert st["kmeans_samples"] == n                       # 256 000 pixels <= 64 x 4096: every pixel is a sample
