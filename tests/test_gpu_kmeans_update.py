"""The order-free centroid update (patolette_amd_set_kmeans_update(1), include/patolette_amd.h): an OPTION next to the
reference's sequential f32 chains, which stay the default and stay bit-exact (test_gpu_parity.py).

What is checked here: the option's centroids are the exactly summed means rounded to f32 -- against numpy in f64 for one iteration
(same assignment on both sides), how close it stays to the reference-faiss goldens and to the oracle's palette (the measured
deviations are printed: the chains' rounding, ~1e-6, plus |x - c| / members for every sample that changes sides), bit-identical
from run to run and across assignment kernels,
and that the setting does not leak (the default is restored and bit-exact again)."""
import ctypes as C

import numpy as np
import pytest

from tests.golden import make_golden as mg
from tests.util import golden

pytestmark = pytest.mark.gpu

dp = C.POINTER(C.c_double)
zp = C.POINTER(C.c_size_t)


def _d(a):
    return a.ctypes.data_as(dp) if a is not None else None


@pytest.fixture
def order_free(gpu):
    before = gpu.patolette_amd_set_kmeans_update(1)
    assert before == 0
    yield gpu
    assert gpu.patolette_amd_set_kmeans_update(0) == 1


def _refine(gpu, x, w, n, cent, k, niter, max_samples):
    c = np.ascontiguousarray(cent.T).reshape(-1).copy()          # planar (k,3)
    assert gpu.patolette_amd_kmeans_refine(_d(x), _d(w), n, _d(c), k, int(niter), int(max_samples)) == 0
    return c.reshape(3, k).T.astype(np.float32)


@pytest.mark.parametrize("n,k,weighted", [(262144, 256, False), (300000, 200, True), (5008, 16, False), (61 * 1147, 61, True), (2000 * 150, 2000, False),
                                          (4096 * 40, 4096, True)])
def test_one_iteration_is_the_exact_mean(gpu, ob, n, k, weighted):
    """One iteration from the same start: the assignment is the reference's (the same kernels make it), so the exact-chain result
    tells which centroids moved, and the order-free centroids must be f32(sum in f64 / f32 count-or-weight) of the same members.
    The members are recovered from the oracle's assignment of the start centroids (nearest centroid in the reference's f32 form)."""
    assert n % k == 0                                         # else faiss subsamples (Clustering.cpp:311-319) and the members below are not all samples
    x, w, cent = mg.km_inputs(n, k, weighted, 5, False)
    want_chain = ob.kmeans_refine(x, w, n, cent, 1, n).astype(np.float32)
    assert gpu.patolette_amd_set_kmeans_update(1) == 0
    try:
        got = _refine(gpu, x, w, n, cent, k, 1, n)
    finally:
        gpu.patolette_amd_set_kmeans_update(0)
    # members by brute force in f64 on the f32 samples: ties and near-ties with the f32 form are possible, so compare centroid-wise
    # with a tolerance that a single moved sample cannot fake for most centroids, and require MOST centroids to match to f32 rounding
    xs = x.reshape(3, n).T.astype(np.float32).astype(np.float64)
    c0 = cent.astype(np.float32).astype(np.float64)
    d = (xs * xs).sum(1)[:, None] - 2.0 * xs @ c0.T + (c0 * c0).sum(1)[None, :]
    a = np.argmin(d, axis=1)
    ws = w.astype(np.float32).astype(np.float64) if weighted else np.ones(n)
    mean = np.zeros((k, 3))
    for j in range(3):
        mean[:, j] = np.bincount(a, weights=xs[:, j] * ws, minlength=k)
    h = np.bincount(a, weights=ws, minlength=k)
    ok = h > 0
    ref = (mean[ok].astype(np.float32) * (np.float32(1) / h[ok].astype(np.float32))[:, None]).astype(np.float32)
    close = np.abs(got[ok].astype(np.float64) - ref.astype(np.float64)).max(1) <= 2.0 ** -22 * np.maximum(1e-30, np.abs(ref).max(1))
    print("centroids equal to the f64 mean to 2 ulp: %d of %d" % (int(close.sum()), int(ok.sum())))
    assert close.mean() > 0.9                                 # the rest: a sample or two assigned differently by the f32 form
    # and within the tolerance of the reference's chains everywhere
    dev = float(np.max(np.abs(got.astype(np.float64) - want_chain.astype(np.float64))))
    print("max deviation from the sequential f32 chains: %.3g" % dev)
    assert dev <= 1e-5


def test_goldens_close_to_the_reference(order_free):
    """Every reference-faiss golden (kmeans_ref.npz) with the order-free update.  One iteration differs by the chains' rounding only
    (~1e-6 of the colour range).  Over several iterations a sample on the border of two cells can land on the other side, and ONE
    sample moves a centroid of m members by |x - c| / m (1e-4 at the ~1000 members of these cases; 1e-6 at the 65 536 of a 4096^2
    image clustered in full): over a few iterations almost all rows stay within 1e-5, the rest within a few times 1 / members; over 32 iterations the
    differences spread to most rows (case 7: up to 2.4e-4).  Planted empty clusters are re-seeded from a
    random draw against the cluster sizes (Clustering.cpp:216-263), which such a sample can redirect: those cases only count rows."""
    gpu = order_free
    g = golden("kmeans_ref.npz")
    for ci, (n, k, niter, max_samples, weighted, seed, plant) in enumerate(g["cases"]):
        n, k = int(n), int(k)
        x, w, cent = mg.km_inputs(n, k, bool(weighted), int(seed), bool(plant))
        got = _refine(gpu, x, w, n, cent, k, niter, max_samples)
        again = _refine(gpu, x, w, n, cent, k, niter, max_samples)
        assert np.array_equal(got.view(np.uint32), again.view(np.uint32)), "case %d: not reproducible" % ci
        ref = g["cent_%d" % ci]
        dev = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max(1)
        print("case %d (n %d k %d it %d w %d plant %d): max dev %.3g, rows beyond 1e-5: %d of %d" % (ci, n, k, niter, weighted, plant, dev.max(), int((dev > 1e-5).sum()), k))
        if niter <= 8:                                            # (32 iterations at 3000 members: ten of thirteen rows end 1e-5 .. 2.4e-4 away)
            assert np.mean(dev <= 1e-5) >= (0.9 if plant else 0.95), "case %d" % ci
        if niter == 1:
            assert dev.max() <= 1e-5, "case %d" % ci
        if not plant:
            assert dev.max() <= 1.0 / (n / k), "case %d" % ci        # a few samples changing sides: |x - c| <= 1 each, over n / k members


@pytest.mark.parametrize("forced", ["lut32", "mid64"])
def test_same_bits_whatever_assignment_kernel(order_free, monkeypatch, forced):
    """The sums are integers: full scan, 32^3-grid and LDS-table assignment kernels (one int or one byte per sample, different
    launch geometries) must end in the same centroid bits."""
    gpu = order_free
    n, k = 400000, 64
    x, w, cent = mg.km_inputs(n, k, True, 9, False)
    base = _refine(gpu, x, w, n, cent, k, 4, n)
    monkeypatch.setenv("PAMD_KM_LUT_MIN", "1")
    if forced == "mid64":
        monkeypatch.setenv("PAMD_KM_G64_MIN", "1")
    got = _refine(gpu, x, w, n, cent, k, 4, n)
    assert np.array_equal(got.view(np.uint32), base.view(np.uint32))


@pytest.mark.parametrize("cs", [1, 2])
def test_end_to_end_palette_within_tolerance_and_map_nearly_identical(gpu, ob, cs):
    """patolette() on a 1024x1024 scene, KMeans over ALL pixels (4096 samples per centroid: the c3full shape scaled down): the
    default (= the oracle, asserted) against the option."""
    import patolette_amd as p
    from tests.util import scene
    W = H = 1024
    n = W * H
    K = 256
    colors = scene(H, W, 31).reshape(-1, 3)

    def run():
        ok, pal, pmap, _ = p.quantize(W, H, colors, K, dither=False, color_space=cs, tile_size=0, kmeans_niter=8, kmeans_max_samples=n)
        assert ok
        return np.array(pal, dtype=np.float64), np.array(pmap)

    pal0, map0 = run()
    ec, pal_o, pmap_o = ob.patolette(W, H, ob.planar(colors), None, K, dither=False, color_space=cs, kmeans_niter=8, kmeans_max_samples=n)
    assert ec == 0 and np.array_equal(map0, pmap_o) and np.allclose(pal0, pal_o, rtol=0, atol=1e-9)
    assert gpu.patolette_amd_set_kmeans_update(1) == 0
    try:
        pal1, map1 = run()
        pal1b, map1b = run()
    finally:
        gpu.patolette_amd_set_kmeans_update(0)
    assert np.array_equal(pal1, pal1b) and np.array_equal(map1, map1b)
    dev = float(np.nanmax(np.abs(pal1 - pal0)))
    differ = int(np.sum(map1 != map0))
    print("palette (sRGB, 0..1) max deviation %.3g; %d of %d map entries differ" % (dev, differ, n))
    # a sample that changes sides moves a centroid of 4096 members by up to 2.4e-5 of the colour range, and sRGB is a non-linear
    # image of the refined centres (measured: 1.4e-4 / 2.0e-4, ~200 map entries of 1 M)
    assert dev <= 1e-3
    assert differ <= n // 1000
    pal2, map2 = run()                                         # the default again: the option has not leaked
    assert np.array_equal(pal2, pal0, equal_nan=True) and np.array_equal(map2, map0)


@pytest.mark.parametrize("weighted", [False, True])
def test_kmeans_over_every_pixel_of_a_16_mp_image(gpu, native, ob, weighted):
    """KMeans over ALL 16.8 M pixels (kmeans_max_samples = N: 4096 chunks of 4096 samples), the size from which the second half of
    the stable counting sort goes through LDS in 4096-sample batches (k_km_scatter_lds): the sorted copy must hold the same records
    in the same places, or the sequential f32 centroid chains (Clustering.cpp:135-204) come out differently.  Against the oracle:
    palette 1e-9, map bit for bit."""
    import ctypes as C
    import os
    w = h = 4096
    n, K = w * h, 256
    L = gpu
    img = L.patolette_amd_malloc(3 * n * 8)
    wt = L.patolette_amd_malloc(n * 8) if weighted else None
    dmap = L.patolette_amd_malloc(n)
    try:
        assert img and dmap and L.patolette_amd_fill_image(img, n, 5) == 0
        if weighted:
            assert wt and L.patolette_amd_fill_weights(wt, n, 5) == 0
        opts = native.QuantizationOptions(False, False, 2, 2, n, False)
        pal = np.zeros((K, 3), dtype=np.float64, order="F")
        code = C.c_int(9)
        L.patolette_amd_device(w, h, img, wt, K, C.byref(opts), pal.ctypes.data_as(C.POINTER(C.c_double)), dmap, 1, C.byref(code))
        assert code.value == 0, native.last_error()
        assert native.last_stats()["kmeans_samples"] == n
        m8 = np.empty(n, dtype=np.uint8)
        assert L.patolette_amd_memcpy_d2h(m8.ctypes.data_as(C.c_void_p), dmap, n) == 0
    finally:
        for q in (img, wt, dmap):
            if q:
                L.patolette_amd_free(q)
    ob.set_threads(os.cpu_count() or 1)
    try:
        ec, pal_o, map_o = ob.patolette(w, h, ob.image(n, 5), ob.weights(n, 5) if weighted else None, K, dither=False, color_space=2,
                                        kmeans_niter=2, kmeans_max_samples=n)
    finally:
        ob.set_threads(1)
    assert ec == 0
    assert np.max(np.abs(pal - pal_o)) <= 1e-9 * np.max(np.abs(pal_o))
    assert int(np.count_nonzero(m8 != map_o.astype(np.uint8))) == 0
