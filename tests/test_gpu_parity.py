"""Parity of the HIP path (through the C ABI of libpatolette_amd.so) against the CPU oracle.

Every test here needs a real MI355X.  Sizes are chosen so the oracle finishes in seconds;
full BASELINE sizes are covered by size-independent properties in test_gpu_properties.py.
Bars: bit-exact for indices (NN map, dither, KMeans assignment -> centroids bit-exact in
f32); 1e-9 relative for f64 palette centres (north_star asks 1e-5); colour conversions
within 1e-12 of the reference arithmetic and >= 98 % bit-identical (device pow is <= 0.51 ulp, not correctly
rounded).
"""
import ctypes as C

import numpy as np
import pytest

from tests.golden import make_golden as mg
from tests.util import golden, match_rows_up_to_permutation

pytestmark = pytest.mark.gpu

dp = C.POINTER(C.c_double)
zp = C.POINTER(C.c_size_t)


def _d(a):
    return a.ctypes.data_as(dp) if a is not None else None


CONV_ID = {"srgb_to_ictcp": 0, "srgb_to_cieluv": 1, "ictcp_to_rec2020": 2, "cieluv_to_rec2020": 3,
           "srgb_to_rec2020": 4, "rec2020_to_srgb": 5}


@pytest.mark.parametrize("name", list(CONV_ID))
def test_convert_matches_oracle(gpu, ob, name):
    n = 200000
    src = ob.image(n, 21)
    if name == "ictcp_to_rec2020":
        src = ob.convert("srgb_to_ictcp", src)
    elif name == "cieluv_to_rec2020":
        src = ob.convert("srgb_to_cieluv", src)
    elif name == "rec2020_to_srgb":
        src = ob.convert("srgb_to_rec2020", src)
    want = ob.convert(name, src)
    got = src.copy()
    assert gpu.patolette_amd_convert(CONV_ID[name], _d(got), n) == 0
    scale = max(1.0, float(np.max(np.abs(want))))
    assert np.max(np.abs(got - want)) <= 1e-12 * scale
    frac = float(np.mean(got == want))
    print("bit-equal fraction %s: %.4f" % (name, frac))
    assert frac > 0.98                           # measured 0.987-0.9985: the rest are last-ulp neighbours from pow / cbrt


def test_convert_reference_golden_edges(gpu):
    g = golden("color_ref.npz")
    n = int(g["n"])
    src = mg.color_inputs(n, int(g["seed"]))
    for name in ("srgb_to_ictcp", "srgb_to_cieluv", "srgb_to_rec2020"):
        got = src.copy()
        assert gpu.patolette_amd_convert(CONV_ID[name], _d(got), n) == 0
        assert np.max(np.abs(got - g[name])) <= 1e-12 * max(1.0, float(np.max(np.abs(g[name]))))
    got = g["srgb_to_cieluv"].copy()
    assert gpu.patolette_amd_convert(6, _d(got), n) == 0          # fused Luv->Rec2020->sRGB->ICtCp chain
    assert np.max(np.abs(got - g["cieluv_to_ictcp"])) <= 1e-11


CLUSTER_CASES = [  # n, K, kind, weighted, colour space applied first
    (65536, 16, "noise", False, "srgb_to_ictcp"),
    (50000, 2, "noise", False, "srgb_to_ictcp"),
    (30000, 64, "blobs", True, "srgb_to_cieluv"),
    (60000, 24, "ramp", False, "srgb_to_ictcp"),
    (120000, 256, "noise", True, "srgb_to_ictcp"),
    (307200, 256, "blobs", False, "srgb_to_ictcp"),
    (2000, 256, "ramp", True, "srgb_to_ictcp"),
    (256, 300, "fewcolors", False, "srgb_to_ictcp"),
    (7, 5, "noise", False, None),
    (1, 4, "noise", False, None),
]


@pytest.mark.parametrize("case", CLUSTER_CASES, ids=lambda c: "%d-%d-%s-%s" % (c[0], c[1], c[2], "w" if c[3] else "u"))
def test_quantize_clusters_matches_oracle(gpu, ob, case):
    n, K, kind, weighted, cs = case
    flat, wt = mg.pipe_input(n, 1, kind, 31, weighted)
    if cs:
        flat = ob.convert(cs, flat)          # identical (oracle-converted) colours go to both sides
    want = ob.quantize_clusters(flat, wt, n, K, want_membership=False)
    centers = np.zeros(3 * K)
    ncl = C.c_size_t(0)
    assert gpu.patolette_amd_quantize_clusters(_d(flat), _d(wt), n, K, _d(centers), C.byref(ncl)) == 0
    assert ncl.value == want["n_clusters"]
    got = centers.reshape(3, K).T[:ncl.value]
    ref = want["centers"][:ncl.value]
    scale = max(1e-30, float(np.nanmax(np.abs(ref))))
    if kind == "fewcolors":                   # collinear clusters: order not defined by the reference (see tests/util.py)
        assert match_rows_up_to_permutation(got, ref, 1e-9 * scale) is not None
        return
    assert np.allclose(got, ref, rtol=0, atol=1e-9 * scale, equal_nan=True), np.nanmax(np.abs(got - ref))


@pytest.mark.parametrize("pruned", [0, 1, 2])
def test_kmeans_bit_exact_vs_reference_faiss_golden(gpu, monkeypatch, pruned):
    """Centroids bit-identical to the reference's faiss; `pruned` = 1 forces the grid-pruned assignment (normally used
    from 2 M samples on) onto these small cases (k < 16 keeps the full scan), 2 also forces the 64^3 grid with the
    four-candidate table held in LDS (normally from 8 M samples on)."""
    if pruned:
        monkeypatch.setenv("PAMD_KM_LUT_MIN", "1")
    if pruned == 2:
        monkeypatch.setenv("PAMD_KM_G64_MIN", "1")
        monkeypatch.setenv("PAMD_KM_LONG_MIN", "1")         # and every centroid through the block-parallel exact chain
    g = golden("kmeans_ref.npz")
    for ci, (n, k, niter, max_samples, weighted, seed, plant) in enumerate(g["cases"]):
        n, k = int(n), int(k)
        x, w, cent = mg.km_inputs(n, k, bool(weighted), int(seed), bool(plant))
        c = np.ascontiguousarray(cent.T).reshape(-1).copy()          # planar (k,3)
        assert gpu.patolette_amd_kmeans_refine(_d(x), _d(w), n, _d(c), k, int(niter), int(max_samples)) == 0
        got = c.reshape(3, k).T.astype(np.float32)
        ref = g["cent_%d" % ci]
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "case %d: %d floats differ" % (
            ci, int(np.sum(got.view(np.uint32) != ref.view(np.uint32))))


@pytest.mark.parametrize("ncol,k,weighted,n", [(20, 256, False, 100000), (5, 200, True, 60000), (40, 64, False, 300000)])
def test_kmeans_many_empty_clusters_every_iteration(gpu, ob, ncol, k, weighted, n):
    """Fewer distinct colours than centroids (a posterised image): dozens to hundreds of clusters come out empty in every iteration
    and are re-seeded by split_clusters (Clustering.cpp:216-263) from ~k draws of mt19937(1234) each -- tens of thousands of
    draws per iteration, i.e. many regenerations of the generator's 624-word block on the wavefront-parallel path."""
    rng = np.random.default_rng(ncol)
    colours = rng.random((ncol, 3))
    pick = rng.integers(0, ncol, size=n)
    pts = colours[pick]
    flat = np.ascontiguousarray(pts.T).reshape(-1).copy()
    w = ob.weights(n, 3) if weighted else None
    cent = pts[rng.choice(n, size=k, replace=False)].copy() + 1e-3 * rng.standard_normal((k, 3))
    want = ob.kmeans_refine(flat, w, n, cent, 4, n)
    c = np.ascontiguousarray(cent.T).reshape(-1).copy()
    assert gpu.patolette_amd_kmeans_refine(_d(flat), _d(w), n, _d(c), k, 4, n) == 0
    got = c.reshape(3, k).T
    assert np.array_equal(got.astype(np.float32).view(np.uint32), want.astype(np.float32).view(np.uint32))


@pytest.mark.parametrize("k,weighted,n,niter", [(12, False, 50000, 40), (64, True, 120000, 60), (256, False, 262144, 25), (9, True, 3000, 50)])
def test_kmeans_fixed_point_and_clean_centroids(gpu, ob, k, weighted, n, niter):
    """Tight, well separated blobs: after a few iterations no sample changes its centroid any more.  The list path then skips the
    update of every centroid whose members did not change (its sequential sums would come out the same) and, once nothing moves
    at all, the remaining iterations -- the oracle runs all of them; the centroids must agree bit for bit after `niter`, and also
    after every smaller count on the way (1, 2, 3, 5, 8: iterations with some clean and some dirty centroids)."""
    rng = np.random.default_rng(k)
    centres = rng.random((k, 3)) * 0.8 + 0.1
    pick = rng.integers(0, k, size=n)
    pts = np.clip(centres[pick] + 0.004 * rng.standard_normal((n, 3)), 0.0, 1.0)
    pts[: n // 7] = rng.random((n // 7, 3))                  # and a share of noise, so that some samples sit on cell borders for a while
    flat = np.ascontiguousarray(pts.T).reshape(-1).copy()
    w = ob.weights(n, 3) if weighted else None
    cent0 = pts[rng.choice(n, size=k, replace=False)].copy()
    for it in (1, 2, 3, 5, 8, niter):
        want = ob.kmeans_refine(flat, w, n, cent0, it, n)
        c = np.ascontiguousarray(cent0.T).reshape(-1).copy()
        assert gpu.patolette_amd_kmeans_refine(_d(flat), _d(w), n, _d(c), k, it, n) == 0
        got = c.reshape(3, k).T
        assert np.array_equal(got.astype(np.float32).view(np.uint32), want.astype(np.float32).view(np.uint32)), it


@pytest.mark.parametrize("k,weighted,n,ncol", [(5000, False, 300000, 0), (4100, True, 70000, 0), (6000, False, 90000, 900), (5000, True, 40000, 0)])
def test_kmeans_palettes_beyond_4096(gpu, ob, k, weighted, n, ncol):
    """The reference refines any palette size (refine.c:77-89, Clustering.cpp:267-554); beyond 4096 entries the stable sort's counters
    and cursors and split_clusters' size table leave LDS for device memory.  Bit-exact centroids against the oracle: subsampled
    (300 000 > k * max_points_per_centroid) and not, weighted, and with `ncol` < k distinct colours so that thousands of clusters
    come out empty in every iteration and are re-seeded."""
    rng = np.random.default_rng(k + n)
    if ncol:
        colours = rng.random((ncol, 3))
        pts = colours[rng.integers(0, ncol, size=n)]
        flat = np.ascontiguousarray(pts.T).reshape(-1).copy()
    else:
        flat = ob.convert("srgb_to_ictcp", ob.image(n, 41))
        pts = flat.reshape(3, n).T
    w = ob.weights(n, 5) if weighted else None
    cent = pts[rng.choice(n, size=k, replace=False)].copy() + 1e-4 * rng.standard_normal((k, 3))
    want = ob.kmeans_refine(flat, w, n, cent, 3, 512 ** 2)
    c = np.ascontiguousarray(cent.T).reshape(-1).copy()
    assert gpu.patolette_amd_kmeans_refine(_d(flat), _d(w), n, _d(c), k, 3, 512 ** 2) == 0
    got = c.reshape(3, k).T
    assert np.array_equal(got.astype(np.float32).view(np.uint32), want.astype(np.float32).view(np.uint32))


def test_end_to_end_5000_colours_with_kmeans(gpu, ob):
    """patolette() with palette_size 5000 and the KMeans refinement on (the stage that used to stop at 4096): palette and map."""
    import patolette_amd as p
    w_, h_ = 640, 480
    n = w_ * h_
    colors = ob.unplanar(ob.image(n, 43), n)
    ok, pal, pmap, msg = p.quantize(w_, h_, colors, 5000, dither=False, color_space=2, tile_size=0, kmeans_niter=3, kmeans_max_samples=512 ** 2)
    ec, pal_o, pmap_o = ob.patolette(w_, h_, ob.planar(colors), None, 5000, dither=False, color_space=2, kmeans_niter=3, kmeans_max_samples=512 ** 2)
    assert ok and ec == 0, msg
    assert np.allclose(pal, pal_o, rtol=0, atol=1e-9)
    assert np.array_equal(pmap, pmap_o)


@pytest.mark.parametrize("cs,k,weighted,g64", [("srgb_to_ictcp", 256, False, False), ("srgb_to_cieluv", 200, True, False), ("srgb_to_ictcp", 61, False, False),
                                               ("srgb_to_ictcp", 256, False, True), ("srgb_to_cieluv", 203, True, True)])
def test_kmeans_pruned_assignment_many_samples(gpu, ob, monkeypatch, cs, k, weighted, g64):
    """2.2 M samples, all clustered: the grid-pruned assignment (automatic at this size) against the oracle's full scans;
    `g64` takes the path of 8 M samples and more (64^3 grid, four-candidate table in LDS, parked overflow samples)."""
    if g64:
        monkeypatch.setenv("PAMD_KM_G64_MIN", "1")
    n = 2200000
    flat = ob.convert(cs, ob.image(n, 31))
    w = ob.weights(n, 31) if weighted else None
    rng = np.random.default_rng(k)
    cent = flat.reshape(3, n).T[rng.choice(n, size=k, replace=False)].copy()
    cent[5] = cent[9]                                       # duplicate centroid: distance ties, one cluster starves
    want = ob.kmeans_refine(flat, w, n, cent, 2, n)
    c = np.ascontiguousarray(cent.T).reshape(-1).copy()
    assert gpu.patolette_amd_kmeans_refine(_d(flat), _d(w), n, _d(c), k, 2, n) == 0
    got = c.reshape(3, k).T
    assert np.array_equal(got.astype(np.float32).view(np.uint32), want.astype(np.float32).view(np.uint32))


def test_u8_batch_equals_separate_u8_calls(gpu):
    """`quantize_u8_batch` (three engines in flight, 3 B/px over PCIe) returns per image exactly what `quantize_u8` returns:
    mixed with / without explicit weights, with the saliency weights (tile_size > 0), K <= 256 and K > 256, RGBA input."""
    import patolette_amd as p
    rng = np.random.default_rng(8)
    for (h, w, ch, K, tile, niter) in [(96, 128, 3, 64, 0, 4), (120, 100, 4, 300, 0, 0), (128, 128, 3, 32, 64, 2)]:
        imgs = [rng.integers(0, 256, size=(h, w, ch), dtype=np.uint8) for _ in range(5)]
        wts = None if tile else [None, 1.0 + rng.random(h * w), None, 1.0 + 3.0 * rng.random(h * w), None]
        got = p.quantize_u8_batch(imgs, K, weights=wts, dither=False, tile_size=tile, kmeans_niter=niter)
        assert len(got) == 5
        for i, g in enumerate(got):
            one = p.quantize_u8(imgs[i], K, weights=None if wts is None else wts[i], dither=False, tile_size=tile, kmeans_niter=niter)
            assert g[0] and one[0]
            for a, b in zip(g[1:5], one[1:5]):
                assert a.dtype == b.dtype and np.array_equal(a, b)
    only = p.quantize_u8_batch(imgs[:2], 16, dither=False, palette_only=True, tile_size=0)
    assert all(o[0] and o[2] is None and o[3] is None and o[1].shape == (16, 3) for o in only)


def test_sharded_batch_takes_8bit_images(gpu):
    """patolette_amd.dist.quantize_batch_sharded (single process here) routes (H, W, 3) uint8 images through the 8-bit batch
    entry and returns the `quantize` tuple: same palette and map as `quantize_u8`."""
    import patolette_amd as p
    from patolette_amd.dist import quantize_batch_sharded
    rng = np.random.default_rng(9)
    h, w, K = 64, 96, 40
    imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(7)]
    out = quantize_batch_sharded(w, h, imgs, K, dither=False, tile_size=0, kmeans_niter=3)
    assert len(out) == 7
    for im, r in zip(imgs, out):
        one = p.quantize_u8(im, K, dither=False, tile_size=0, kmeans_niter=3)
        assert r[0] and np.array_equal(r[1], one[4]) and np.array_equal(r[2], one[2].reshape(-1))


@pytest.mark.parametrize("weighted", [False, True])
def test_kmeans_long_chains_with_ties_and_binade_changes(gpu, ob, weighted):
    """Clusters of 8192 samples and more sum their members as exact integer sums per 1024-sample block while the f32
    accumulator keeps its binade; exact ties of the rounding (the parity of the running sum decides) and blocks that could
    leave the binade are replayed in order.  Samples taken from a handful of dyadic values make ties the rule rather than
    the exception, mixed signs make the accumulator wander across binades: centroids bit-identical to the oracle's
    sequential sums."""
    n, k = 400000, 16
    rng = np.random.default_rng(17)
    vals = np.array([0.5, 0.25, 0.75, -0.125, 0.0625, 1.0, -0.5, 0.3125])
    pts = vals[rng.integers(0, len(vals), size=(n, 3))] + (rng.integers(0, 4, size=(n, 3)) == 0) * rng.standard_normal((n, 3)) * 1e-3
    flat = np.ascontiguousarray(pts.T).reshape(-1)
    w = (1.0 + rng.integers(0, 4, size=n) * 0.5) if weighted else None
    cent = pts[rng.choice(n, size=k, replace=False)].copy()
    want = ob.kmeans_refine(flat, w, n, cent, 3, n)
    c = np.ascontiguousarray(cent.T).reshape(-1).copy()
    assert gpu.patolette_amd_kmeans_refine(_d(flat), _d(w), n, _d(c), k, 3, n) == 0
    got = c.reshape(3, k).T
    assert np.array_equal(got.astype(np.float32).view(np.uint32), want.astype(np.float32).view(np.uint32))


@pytest.mark.parametrize("k,weighted", [(64, False), (256, True)])
def test_kmeans_lds_table_path_with_exact_distance_ties(gpu, ob, monkeypatch, k, weighted):
    """The LDS-table assignment evaluates a cell's four candidates as 64-bit {distance : index} keys and hands every doubtful
    sample (another index at the same smallest distance, a negative rounded distance) to the reference's exact procedure.
    Samples on a coarse dyadic lattice around the origin, centroids started ON lattice points: coincident samples
    (distance 0), equidistant centroids in different and in equal SIMD lanes, and sums with no cancellation at all are the
    rule here; centroids bit-identical to the oracle's full scans."""
    monkeypatch.setenv("PAMD_KM_LUT_MIN", "1")
    monkeypatch.setenv("PAMD_KM_G64_MIN", "1")
    n = 600000
    rng = np.random.default_rng(100 + k)
    lattice = rng.integers(-8, 9, size=(n, 3)) / 16.0
    jitter = (rng.integers(0, 3, size=(n, 1)) == 0) * rng.standard_normal((n, 3)) * 2e-3
    pts = (lattice + jitter).astype(np.float32).astype(np.float64)
    flat = np.ascontiguousarray(pts.T).reshape(-1)
    w = (1.0 + rng.integers(0, 3, size=n) * 0.25) if weighted else None
    cent = (rng.integers(-8, 9, size=(k, 3)) / 16.0 + (rng.integers(0, 2, size=(k, 1)) * rng.standard_normal((k, 3)) * 1e-2))
    cent[3] = cent[11]                                      # equal centroids in one SIMD lane (3 = 11 mod 8) ...
    cent[4] = cent[13]                                      # ... and in different lanes
    want = ob.kmeans_refine(flat, w, n, cent, 3, n)
    c = np.ascontiguousarray(cent.T).reshape(-1).copy()
    assert gpu.patolette_amd_kmeans_refine(_d(flat), _d(w), n, _d(c), k, 3, n) == 0
    got = c.reshape(3, k).T
    assert np.array_equal(got.astype(np.float32).view(np.uint32), want.astype(np.float32).view(np.uint32))


def test_nn_map_bit_exact(gpu, ob):
    for n, k, seed in [(100000, 256, 1), (5000, 7, 2), (333, 1, 3), (70000, 300, 4)]:
        flat = ob.convert("srgb_to_ictcp", ob.image(n, seed))
        pal = ob.convert("srgb_to_ictcp", ob.image(k, 50 + seed)).reshape(3, k).T.copy()
        if k > 4:
            pal[3] = pal[1]                   # exact duplicate -> tie -> lowest index must win
        want = ob.nn_map(flat, n, pal)
        got = np.zeros(n, dtype=np.uintp)
        p = np.ascontiguousarray(pal.T).reshape(-1)
        assert gpu.patolette_amd_nn_map(_d(flat), n, _d(p), k, got.ctypes.data_as(zp)) == 0
        assert np.array_equal(got, want)
        assert not np.any(got == 3) or k <= 4


def test_nn_map_pruned_path_edge_cases(gpu, ob):
    """The grid-pruned kernel must stay bit-exact: crowded palettes (list overflow -> full scan),
    a degenerate axis, duplicate entries, palette entries far outside the pixel box."""
    n = 200000
    base = ob.convert("srgb_to_ictcp", ob.image(n, 12))
    cases = []
    k = 256
    crowded = np.tile(np.array([[0.3, 0.01, -0.02]]), (k, 1)) + 1e-6 * ob.image(k, 5).reshape(3, k).T   # all within 1e-6
    cases.append((base, crowded))
    flat2 = base.copy(); flat2[n:2 * n] = 0.0125                                   # constant plane
    cases.append((flat2, ob.convert("srgb_to_ictcp", ob.image(k, 6)).reshape(3, k).T.copy()))
    far = ob.convert("srgb_to_ictcp", ob.image(64, 7)).reshape(3, 64).T.copy()
    far[::2] += 5.0                                                                 # half the palette far away
    far[10] = far[12]
    cases.append((base, far))
    small_k = ob.convert("srgb_to_ictcp", ob.image(9, 8)).reshape(3, 9).T.copy()
    cases.append((base, small_k))
    for flat, pal in cases:
        kk = pal.shape[0]
        want = ob.nn_map(flat, n, pal)
        got = np.zeros(n, dtype=np.uintp)
        assert gpu.patolette_amd_nn_map(_d(flat), n, _d(np.ascontiguousarray(pal.T).reshape(-1)), kk, got.ctypes.data_as(zp)) == 0
        assert np.array_equal(got, want), int(np.sum(got != want))


def test_nn_map_lds_table_path_is_exact(gpu, ob):
    """From 4 Mpx on (and K <= 256) the map kernel keeps a 32^3 table of four-candidate entries in LDS and sends the
    pixels of crowded cells through a per-wavefront queue: every pixel of a 4.2 Mpx image against the oracle's brute
    force, for a quantiser's palette with duplicates, a palette crowded into one corner (every pixel overflows), and a
    constant plane with half of the palette far outside the pixel box."""
    n = (1 << 22) + 777
    base = ob.convert("srgb_to_ictcp", ob.image(n, 21))
    small = ob.convert("srgb_to_ictcp", ob.image(256 * 256, 22))
    pal = ob.quantize_clusters(small, None, 256 * 256, 256, want_membership=False)["centers"].copy()
    pal[200] = pal[17]                                                             # exact duplicate: the lower index must win
    crowded = np.tile(np.array([[0.05, 0.0, 0.01]]), (256, 1)) + 1e-5 * ob.image(256, 5).reshape(3, 256).T
    flat2 = base.copy(); flat2[n:2 * n] = 0.0125
    far = ob.convert("srgb_to_ictcp", ob.image(200, 7)).reshape(3, 200).T.copy()
    far[::2] += 5.0
    for flat, p in ((base, pal), (base, crowded), (flat2, far)):
        kk = p.shape[0]
        want = ob.nn_map(flat, n, p)
        got = np.zeros(n, dtype=np.uintp)
        assert gpu.patolette_amd_nn_map(_d(flat), n, _d(np.ascontiguousarray(p.T).reshape(-1)), kk, got.ctypes.data_as(zp)) == 0
        assert np.array_equal(got, want), int(np.sum(got != want))


@pytest.mark.parametrize("wh", [(64, 64), (37, 23), (5, 40), (1, 9), (130, 70), (256, 3)])
def test_dither_bit_exact(gpu, ob, wh):
    w, h = wh
    n = w * h
    for k in (5, 64, 100, 128, 130, 256):           # one, two and four entries per lane (100 and 130: lanes with padding entries)
        flat = ob.convert("srgb_to_rec2020", ob.image(n, 7))
        pal = ob.convert("srgb_to_rec2020", ob.image(k, 9)).reshape(3, k).T.copy()
        want = ob.dither(flat, w, h, pal)
        got = np.zeros(n, dtype=np.uintp)
        p = np.ascontiguousarray(pal.T).reshape(-1)
        assert gpu.patolette_amd_dither(_d(flat), w, h, _d(p), k, got.ctypes.data_as(zp)) == 0
        assert np.array_equal(got, want), "%dx%d k=%d: %d mismatches" % (w, h, k, int(np.sum(got != want)))


@pytest.mark.parametrize("k", [700, 3300, 5000])
def test_dither_large_palettes(gpu, ob, k):
    """Palettes beyond 256 entries (the general per-lane loop over LDS tables) and beyond 3200, where the two tables no longer
    fit LDS and sit in global memory (the reference takes any palette size, riemersma.c:437-459)."""
    w, h = 96, 50
    n = w * h
    flat = ob.convert("srgb_to_rec2020", ob.image(n, 17))
    pal = ob.convert("srgb_to_rec2020", ob.image(k, 19)).reshape(3, k).T.copy()
    want = ob.dither(flat, w, h, pal)
    got = np.zeros(n, dtype=np.uintp)
    p = np.ascontiguousarray(pal.T).reshape(-1)
    assert gpu.patolette_amd_dither(_d(flat), w, h, _d(p), k, got.ctypes.data_as(zp)) == 0
    assert np.array_equal(got, want), "k=%d: %d mismatches" % (k, int(np.sum(got != want)))


def test_dither_1x1_leaves_map_untouched(gpu, ob):
    flat = ob.image(1, 3)
    pal = np.array([[0.1, 0.2, 0.3], [0.5, 0.5, 0.5]])
    got = np.full(1, 77, dtype=np.uintp)
    assert gpu.patolette_amd_dither(_d(flat), 1, 1, _d(np.ascontiguousarray(pal.T).reshape(-1)), 2, got.ctypes.data_as(zp)) == 0
    assert got[0] == 77                        # riemersma.c:452-456: level 0 visits nothing


def _run_native(native, w, h, flat, wt, K, **kw):
    L = native.lib()
    opts = native.QuantizationOptions(kw.get("dither", True), kw.get("palette_only", False), kw.get("color_space", 2),
                                      kw.get("kmeans_niter", 32), kw.get("kmeans_max_samples", 512 ** 2), False)
    pal = np.zeros((K, 3), dtype=np.float64, order="F")
    n = w * h
    pmap = np.zeros(n, dtype=np.uintp)
    code = C.c_int(99)
    L.patolette(w, h, _d(flat), _d(wt), K, C.byref(opts), pal.ctypes.data_as(dp), pmap.ctypes.data_as(zp), C.byref(code))
    return code.value, pal, pmap


@pytest.mark.parametrize("ci", range(len(mg.PIPE_CASES)))
def test_end_to_end_matches_golden(gpu, native, ci):
    """patolette() on the GPU vs the committed end-to-end vectors (incl. BASELINE config 1)."""
    g = golden("pipeline_oracle.npz")
    w, h, K, cs, niter, dither, weighted, kind, seed = mg.PIPE_CASES[ci]
    flat, wt = mg.pipe_input(w, h, kind, seed, weighted)
    ec, pal, pmap = _run_native(native, w, h, flat, wt, K, dither=dither, color_space=cs, kmeans_niter=niter,
                                kmeans_max_samples=65536)
    assert ec == int(g["ec_%d" % ci])
    ref_pal, ref_map = g["pal_%d" % ci], g["map_%d" % ci]
    assert np.array_equal(pal == -1, ref_pal == -1)
    if cs == 0 and not dither:
        return                                  # reference quirk: sRGB + NN yields a garbage palette
    tol = 1e-5 if niter > 0 else 1e-9           # KMeans centroids are f32
    if kind == "fewcolors":                     # collinear clusters: palette order ambiguous, compare up to permutation
        perm = match_rows_up_to_permutation(pal, ref_pal, tol)
        assert perm is not None
        inv = np.empty_like(perm)
        inv[perm] = np.arange(len(perm))
        assert np.array_equal(inv[pmap.astype(np.int64)].astype(np.uint16), ref_map)
        return
    assert np.allclose(pal, ref_pal, rtol=0, atol=tol, equal_nan=True), float(np.nanmax(np.abs(pal - ref_pal)))
    if w * h == 1 and dither:
        return
    assert np.array_equal(pmap.astype(np.uint16), ref_map), "%d map mismatches" % int(np.sum(pmap.astype(np.uint16) != ref_map))


def test_python_quantize_tuple_contract(gpu, ob):
    import patolette_amd as p
    w, h, K = 80, 50, 12
    n = w * h
    flat = ob.image(n, 2)
    colors = flat.reshape(3, n).T
    ok, pal, pmap, msg = p.quantize(w, h, colors, K, dither=False, tile_size=0, kmeans_niter=0)
    assert ok is True and msg == "Quantization successful."
    assert pal.shape == (K, 3) and pal.dtype == np.float64 and pal.flags.f_contiguous
    assert pmap.shape == (n,) and pmap.dtype == np.uintp
    ec, pal_o, pmap_o = ob.patolette(w, h, flat, None, K, dither=False, kmeans_niter=0)
    assert np.allclose(pal, pal_o, atol=1e-9) and np.array_equal(pmap, pmap_o)
    ok, pal2, pmap2, msg = p.quantize(w, h, colors, K, palette_only=True, tile_size=0, kmeans_niter=0)
    assert ok and pmap2 is None
    ok, pal3, pmap3, msg = p.quantize(w, h, colors, 0, tile_size=0)
    assert (ok, pal3, pmap3, msg) == (False, None, None, "Palette size should be greater than 0.")
    res = p.quantize_batch(w, h, [colors, colors[::-1].copy()], K, dither=False, tile_size=0, kmeans_niter=0)
    assert np.array_equal(res[0][2], pmap) and res[1][0]


def test_batch_equals_individual_calls(gpu, ob):
    import patolette_amd as p
    w, h, K, count = 96, 80, 32, 7
    n = w * h
    imgs = [ob.image(n, 200 + i).reshape(3, n).T.copy() for i in range(count)]
    wts = [ob.weights(n, 200 + i) if i % 2 else None for i in range(count)]
    # images without explicit weights take the saliency-derived ones (tile_size > 0), like quantize()
    batch = p.quantize_batch(w, h, imgs, K, weights=wts, dither=False, tile_size=64, kmeans_niter=4, kmeans_max_samples=65536)
    for i in range(count):
        one = p.quantize(w, h, imgs[i], K, dither=False, tile_size=64, kmeans_niter=4, kmeans_max_samples=65536, weights=wts[i])
        assert batch[i][0] and one[0]
        assert np.array_equal(batch[i][1], one[1]) and np.array_equal(batch[i][2], one[2])


def test_batch_of_large_images_six_in_flight(gpu, ob):
    """Eight images of 1536 x 1408 (2.16 Mpx: above the size from which the host entry uploads in pieces and converts behind the
    copies on a second stream, and from which KMeans subsamples on the helper thread) through the batch entry -- six engines in
    flight, each with its own streams, helper thread and workspace -- against separate calls, one of them against the oracle."""
    import os
    import patolette_amd as p
    w, h, K, count = 1536, 1408, 64, 8
    n = w * h
    imgs = [np.asfortranarray(ob.unplanar(ob.image(n, 400 + i), n)) if i % 2 else ob.unplanar(ob.image(n, 400 + i), n) for i in range(count)]   # planar and row-major
    wts = [ob.weights(n, 400 + i) if i % 3 == 0 else None for i in range(count)]
    kw = dict(dither=False, tile_size=0, kmeans_niter=3, kmeans_max_samples=512 ** 2)
    batch = p.quantize_batch(w, h, imgs, K, weights=wts, **kw)
    for i in range(count):
        one = p.quantize(w, h, imgs[i], K, weights=wts[i], **kw)
        assert batch[i][0] and one[0]
        assert np.array_equal(batch[i][1], one[1]) and np.array_equal(batch[i][2], one[2]), i
    ob.set_threads(os.cpu_count() or 1)
    try:
        ec, pal_o, map_o = ob.patolette(w, h, ob.planar(np.ascontiguousarray(imgs[3])), wts[3], K, dither=False, color_space=2, kmeans_niter=3, kmeans_max_samples=512 ** 2)
    finally:
        ob.set_threads(1)
    assert ec == 0 and np.allclose(batch[3][1], pal_o, rtol=0, atol=1e-9) and np.array_equal(batch[3][2], map_o)


@pytest.mark.parametrize("channels,K,dither,cs", [(3, 64, False, 2), (4, 64, True, 1), (3, 300, False, 0), (3, 16, True, 2)])
def test_u8_adaptor_equals_by_hand_steps(gpu, ob, channels, K, dither, cs):
    """quantize_u8 == what README.md:147-194 does by hand around quantize(), and == the oracle fed with img/255."""
    import patolette_amd as p
    w, h = 83, 61
    n = w * h
    rng = np.random.default_rng(5 + K)
    img = rng.integers(0, 256, size=(h, w, channels), dtype=np.uint8)
    wts = ob.weights(n, 9) if dither else None
    ok, pal8, pmap, quant, pal, msg = p.quantize_u8(img, K, dither=dither, color_space=cs, tile_size=0, kmeans_niter=3,
                                                    kmeans_max_samples=4096, weights=wts)
    assert ok and msg == "Quantization successful."
    colors = img[:, :, :3].reshape(-1, 3).astype(np.float64)
    colors /= 255
    ok2, pal_f, pmap_f, _ = p.quantize(w, h, colors, K, dither=dither, color_space=cs, tile_size=0, kmeans_niter=3,
                                       kmeans_max_samples=4096, weights=wts)
    assert ok2
    assert np.array_equal(pal, pal_f)
    assert np.array_equal(pmap.reshape(-1).astype(np.uintp), pmap_f)
    by_hand = np.clip(pal_f * 255, 0, 255).astype(np.uint8)
    assert pal8.dtype == np.uint8 and np.array_equal(pal8, by_hand)
    assert np.array_equal(quant, by_hand[pmap_f].reshape(h, w, 3))
    assert pmap.dtype == (np.uint8 if K <= 256 else np.uint16)
    ec, pal_o, pmap_o = ob.patolette(w, h, ob.planar(colors), wts, K, dither=dither, color_space=cs, kmeans_niter=3,
                                     kmeans_max_samples=4096)
    assert ec == 0 and np.array_equal(pmap_f, pmap_o)
    assert np.allclose(pal_f, pal_o, rtol=0, atol=1e-9)
    # palette only / no reconstructed image
    ok3, pal8b, pm3, q3, _, _ = p.quantize_u8(img, K, palette_only=True, color_space=cs, tile_size=0, kmeans_niter=3,
                                              kmeans_max_samples=4096, dither=dither, weights=wts)
    # palette_only leaves the palette in the quantisation space, as the reference does (patolette.c:267 skips the
    # back-conversion together with the map): the adaptor reports exactly what the by-hand steps would
    ok4, pal_po, _, _ = p.quantize(w, h, colors, K, palette_only=True, color_space=cs, tile_size=0, kmeans_niter=3,
                                   kmeans_max_samples=4096, dither=dither, weights=wts)
    assert ok3 and ok4 and pm3 is None and q3 is None
    assert np.array_equal(pal8b, np.clip(pal_po * 255, 0, 255).astype(np.uint8))
    # palette wanted in sRGB but no map / image: want_quantized=False and the map dropped by the caller still converts
    ok5, pal8c, pm5, q5, _, _ = p.quantize_u8(img, K, color_space=cs, tile_size=0, kmeans_niter=3, kmeans_max_samples=4096,
                                              dither=dither, weights=wts, want_quantized=False)
    assert ok5 and q5 is None and np.array_equal(pal8c, pal8) and np.array_equal(pm5, pmap)


@pytest.mark.parametrize("y", [2.4, 1 / 2.4, 0.1593017578125, 78.84375, 1 / 0.1593017578125, 1 / 78.84375, 1.0 / 3.0, 3.0])
def test_device_pow_within_one_ulp_of_libm(gpu, native, y):
    """The conversions' pow(): never more than one ulp from the host libm (correctly rounded in practice), >= 99 % identical."""
    rng = np.random.default_rng(3)
    n = 400000
    u = rng.random(n)
    x = np.concatenate([u[: n // 4], np.exp((u[n // 4: n // 2] - 0.6) * 20), 0.5 + u[n // 2: 3 * n // 4], u[3 * n // 4:] * 1e4,
                        [0.0, 1.0, 2.0, 0.5, 1e-300, 4e-320, 1e300]])
    out = np.zeros_like(x)
    assert native.lib().patolette_amd_pow(_d(x), y, _d(out), x.size) == 0
    import math

    def libm_pow(v):                                           # glibc's scalar pow (numpy's vectorised power is ~1 ulp)
        try:
            return math.pow(v, y)
        except OverflowError:
            return math.inf
    want = np.array([libm_pow(v) for v in x])
    d = np.abs(out.view(np.int64) - want.view(np.int64))
    finite = np.isfinite(want) & (np.abs(want) > 1e-300)       # gradual underflow: not compared bit for bit
    assert d[finite].max() <= 1
    assert np.mean(d[finite] == 0) >= 0.99
    big = ~np.isfinite(want)
    assert np.array_equal(out[big], want[big])
    neg = np.array([-1.0, -0.25, np.nan])
    o2 = np.zeros(3)
    native.lib().patolette_amd_pow(_d(neg), y, _d(o2), 3)
    assert np.all(np.isnan(o2)) or float(y).is_integer()


@pytest.mark.parametrize("tile_size", [0, 48])
def test_row_major_and_planar_inputs_agree(gpu, ob, tile_size):
    """quantize() hands a C-ordered (N,3) array over row-major and an F-ordered one planar: same results either way."""
    import patolette_amd as p
    from tests.util import scene
    rows, cols, K = 70, 90, 20
    img = scene(rows, cols, 8)
    c_order = np.ascontiguousarray(img.reshape(-1, 3))
    f_order = np.asfortranarray(c_order)
    assert c_order.flags.c_contiguous and f_order.flags.f_contiguous and not f_order.flags.c_contiguous
    a = p.quantize(cols, rows, c_order, K, dither=True, color_space=1, tile_size=tile_size, kmeans_niter=3, kmeans_max_samples=4096)
    b = p.quantize(cols, rows, f_order, K, dither=True, color_space=1, tile_size=tile_size, kmeans_niter=3, kmeans_max_samples=4096)
    c = p.quantize(cols, rows, c_order.astype(np.float32), K, dither=True, color_space=1, tile_size=tile_size, kmeans_niter=3,
                   kmeans_max_samples=4096)          # another dtype: cast, then row-major
    assert a[0] and b[0] and c[0]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert c[1].shape == a[1].shape


def test_subsample_list_with_and_without_the_cache(gpu, native, ob):
    """A subsampled KMeans refinement (400 x 300 pixels, 65 536 samples) with the device-side cache of the subsample list on
    (default), off (every call makes the list again on the helper thread) and after a different image size has replaced the
    cached list: always the oracle's palette and map."""
    import patolette_amd as p
    w, h, K = 400, 300, 32
    n = w * h
    flat = ob.image(n, 71)
    colors = ob.unplanar(flat, n)
    ec, pal_o, pmap_o = ob.patolette(w, h, flat, None, K, dither=False, color_space=2, kmeans_niter=3, kmeans_max_samples=65536)
    small = ob.unplanar(ob.image(350 * 260, 72), 350 * 260)
    L = native.lib()
    prev = L.patolette_amd_set_subsample_cache(1)
    try:
        for cache, between in ((1, False), (1, False), (0, False), (0, False), (1, True), (0, True)):
            L.patolette_amd_set_subsample_cache(cache)
            if between:                                      # another size in between: the cached list belongs to other dimensions
                assert p.quantize(350, 260, small, K, dither=False, tile_size=0, kmeans_niter=1, kmeans_max_samples=65536)[0]
            ok, pal, pmap, msg = p.quantize(w, h, colors, K, dither=False, color_space=2, tile_size=0, kmeans_niter=3, kmeans_max_samples=65536)
            assert ok and ec == 0, msg
            assert np.allclose(pal, pal_o, rtol=0, atol=1e-9), (cache, between)
            assert np.array_equal(pmap, pmap_o), (cache, between)
    finally:
        L.patolette_amd_set_subsample_cache(prev)


@pytest.mark.parametrize("rows_major,weighted,cs,n_side", [(False, False, 2, (300, 420)), (True, False, 1, (257, 391)), (False, True, 2, (128, 96)), (True, True, 0, (64, 33))])
def test_chunked_upload_with_overlapped_conversion(gpu, ob, monkeypatch, rows_major, weighted, cs, n_side):
    """The host entry uploads large images in eight chunks and converts each behind the next one's copy (second stream, the
    statistics accumulated over the launches); forced here onto small images -- chunk boundaries not multiples of anything,
    planar and row-major sources, explicit weights -- the result must be the oracle's as for the single-copy path."""
    import patolette_amd as p
    monkeypatch.setenv("PAMD_UPLOAD_CHUNK_MIN", "1000")
    h, w = n_side
    n = w * h
    flat = ob.image(n, 61)
    colors = ob.unplanar(flat, n)                            # C-contiguous (N,3): the row-major entry
    if not rows_major:
        colors = np.asfortranarray(colors)                   # planar: the drop-in layout
    wts = ob.weights(n, 61) if weighted else None
    ok, pal, pmap, msg = p.quantize(w, h, colors, 48, dither=False, color_space=cs, tile_size=0, kmeans_niter=2, kmeans_max_samples=65536, weights=wts)
    ec, pal_o, pmap_o = ob.patolette(w, h, flat, wts, 48, dither=False, color_space=cs, kmeans_niter=2, kmeans_max_samples=65536)
    assert ok and ec == 0, msg
    assert np.allclose(pal, pal_o, rtol=0, atol=1e-9)
    assert np.array_equal(pmap, pmap_o)


@pytest.mark.parametrize("channels,weighted,hw", [(3, False, (61, 83)), (4, True, (97, 45)), (3, True, (33, 129))])
def test_u8_entry_chunked_upload_with_overlapped_conversion(gpu, ob, monkeypatch, channels, weighted, hw):
    """patolette_amd_u8 uploads large 8-bit images in chunks and converts each behind the next one's copy (run_u8: second stream,
    the statistics started by chunk 0 only); forced here onto small images of odd sizes, 3 and 4 bytes per pixel, with and without
    explicit weights: identical to the single-copy path and equal to the oracle fed with the by-hand conversion (README.md:156-158)."""
    import patolette_amd as p
    h, w = hw
    rng = np.random.default_rng(5 + channels)
    img = rng.integers(0, 256, size=(h, w, channels), dtype=np.uint8)
    wts = (1.0 + 3.0 * rng.random(h * w)) if weighted else None
    kw = dict(dither=False, color_space=2, tile_size=0, kmeans_niter=2, kmeans_max_samples=65536, weights=wts)
    ref = p.quantize_u8(img, 40, **kw)
    monkeypatch.setenv("PAMD_UPLOAD_CHUNK_MIN", "1000")
    got = p.quantize_u8(img, 40, **kw)
    assert ref[0] and got[0], (ref[5], got[5])
    for a, b in zip(ref[1:5], got[1:5]):
        assert np.array_equal(a, b)
    flat = ob.planar(img[:, :, :3].reshape(-1, 3).astype(np.float64) / 255)
    ec, pal_o, map_o = ob.patolette(w, h, flat, wts, 40, dither=False, color_space=2, kmeans_niter=2, kmeans_max_samples=65536)
    assert ec == 0 and np.allclose(got[4], pal_o, rtol=0, atol=1e-9)
    assert np.array_equal(got[2].reshape(-1), map_o)


def test_u8_adaptor_accepts_torch_cuda_tensor(gpu):
    """A torch CUDA uint8 tensor goes through the device entry point: same results as the numpy path, outputs stay in
    HBM.  Own process: torch has to load its HIP runtime before libpatolette_amd.so does (as bench.py --gpus N does)."""
    import subprocess
    import sys
    from tests.util import ROOT
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import torch
assert torch.cuda.is_available()
import patolette_amd as p
from tests.util import scene
img = np.round(scene(96, 128, 2) * 255).astype(np.uint8)
ref = p.quantize_u8(img, 32, dither=False, tile_size=32, kmeans_niter=3, kmeans_max_samples=4096)
got = p.quantize_u8(torch.from_numpy(img).cuda(), 32, dither=False, tile_size=32, kmeans_niter=3, kmeans_max_samples=4096)
assert ref[0] and got[0]
assert got[2].is_cuda and got[3].is_cuda
assert np.array_equal(got[1], ref[1]) and np.array_equal(got[4], ref[4])
assert np.array_equal(got[2].cpu().numpy(), ref[2]) and np.array_equal(got[3].cpu().numpy(), ref[3])
print("TORCH-PATH-OK")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "TORCH-PATH-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_verbose_prints_the_reference_progress_lines(gpu, ob, capfd):
    """verbose=True prints the stage lines patolette.c:209-303 / patolette.pyx:409 print (faiss' own chatter excepted)."""
    import ctypes
    import patolette_amd as p
    from tests.util import scene
    rows, cols = 40, 50
    colors = scene(rows, cols, 3).reshape(-1, 3)
    ok, *_ = p.quantize(cols, rows, colors, 8, dither=False, tile_size=16, kmeans_niter=2, kmeans_max_samples=1024, verbose=True)
    assert ok
    ctypes.CDLL(None).fflush(None)
    out = capfd.readouterr().out
    for line in ("patolette ======== Generating saliency map", "patolette ======== Palette generation ",
                 "patolette ======== Base cluster count: ", "patolette ======== KMeans refinement", "patolette ======== NN mapping"):
        assert line in out, (line, out)
    p.quantize(cols, rows, colors, 8, dither=True, tile_size=0, kmeans_niter=0, verbose=True)
    ctypes.CDLL(None).fflush(None)
    assert "patolette ======== Dithering" in capfd.readouterr().out


@pytest.mark.parametrize("rows,cols,K,cs,dither,niter", [(768, 1024, 256, 2, False, 4), (300, 420, 96, 1, True, 2), (512, 512, 17, 0, False, 3)])
def test_photograph_like_image_end_to_end(gpu, ob, rows, cols, K, cs, dither, niter):
    """Smooth gradients + flat blobs + mild noise (tight clusters, near-degenerate splits): palette and map against the
    oracle, the map bit for bit."""
    import patolette_amd as p
    from tests.util import scene
    img = scene(rows, cols, 17)
    colors = img.reshape(-1, 3)
    ok, pal, pmap, _ = p.quantize(cols, rows, colors, K, dither=dither, color_space=cs, tile_size=0, kmeans_niter=niter,
                                  kmeans_max_samples=65536)
    ec, pal_o, pmap_o = ob.patolette(cols, rows, ob.planar(colors), None, K, dither=dither, color_space=cs, kmeans_niter=niter,
                                     kmeans_max_samples=65536)
    assert ok and ec == 0
    assert np.allclose(pal, pal_o, rtol=0, atol=1e-9)
    assert np.array_equal(pmap, pmap_o)


@pytest.mark.parametrize("rows,cols,K,cs,kind", [(1200, 1600, 256, 1, "noise"), (1024, 1536, 200, 2, "scene")])
def test_large_weighted_image_end_to_end(gpu, ob, rows, cols, K, cs, kind):
    """1.6-1.9 Mpx with weights: the sizes from which the global quantiser's histogram sums in fixed point (2^18 pixels) and the
    partition kernels walk many tiles per block, weighted variants -- palette and map against the oracle, the map bit for bit."""
    import patolette_amd as p
    from tests.util import scene
    n = rows * cols
    colors = ob.unplanar(ob.image(n, 23), n) if kind == "noise" else scene(rows, cols, 29).reshape(-1, 3)
    wts = ob.weights(n, 23)
    ok, pal, pmap, _ = p.quantize(cols, rows, colors, K, dither=False, color_space=cs, tile_size=0, kmeans_niter=2,
                                  kmeans_max_samples=512 ** 2, weights=wts)
    ec, pal_o, pmap_o = ob.patolette(cols, rows, ob.planar(colors), wts, K, dither=False, color_space=cs, kmeans_niter=2,
                                     kmeans_max_samples=512 ** 2)
    assert ok and ec == 0
    assert np.allclose(pal, pal_o, rtol=0, atol=1e-9)
    assert np.array_equal(pmap, pmap_o)


@pytest.mark.parametrize("entry", ["f64", "u8"])
def test_chunked_upload_with_derived_weights(gpu, monkeypatch, entry):
    """With the saliency weights derived on the device (tile_size > 0, the binding's default) the image still goes up in pieces with the
    conversion of each behind the next one's copy: the saliency stage reads the sRGB source, which the conversion leaves alone, and the
    weights' plane is reserved up front.  Forced onto a small image: identical to the single-copy path, both entries."""
    import patolette_amd as p
    rng = np.random.default_rng(41)
    h, w, K = 83, 97, 24
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)

    def call():
        if entry == "u8":
            r = p.quantize_u8(img, K, dither=False, color_space=2, tile_size=32, kmeans_niter=2, kmeans_max_samples=65536)
            assert r[0], r[5]
            return r[1:5]
        ok, pal, pmap, msg = p.quantize(w, h, img.reshape(-1, 3).astype(np.float64) / 255, K, dither=False, color_space=1, tile_size=32,
                                        kmeans_niter=2, kmeans_max_samples=65536)
        assert ok, msg
        return pal, pmap
    ref = call()
    monkeypatch.setenv("PAMD_UPLOAD_CHUNK_MIN", "1000")
    got = call()
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
