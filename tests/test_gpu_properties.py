"""Size-independent properties of the HIP path at BASELINE.json's full sizes (configs[1..3]),
where the CPU oracle would take minutes.  Needs a real MI355X."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

dp = C.POINTER(C.c_double)


class Dev:
    """A synthetic image resident in HBM (splitmix64 generator)."""

    def __init__(self, L, n, seed, weighted=False):
        self.L, self.n = L, n
        self.img = L.patolette_amd_malloc(3 * n * 8)
        assert self.img and L.patolette_amd_fill_image(self.img, n, seed) == 0
        self.w = None
        if weighted:
            self.w = L.patolette_amd_malloc(n * 8)
            assert self.w and L.patolette_amd_fill_weights(self.w, n, seed) == 0
        self.map = L.patolette_amd_malloc(n)

    def free(self):
        for p in (self.img, self.w, self.map):
            if p:
                self.L.patolette_amd_free(p)

    def host_image(self):
        out = np.empty(3 * self.n)
        assert self.L.patolette_amd_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.img, out.nbytes) == 0
        return out

    def host_map(self):
        out = np.empty(self.n, dtype=np.uint8)
        assert self.L.patolette_amd_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.map, out.nbytes) == 0
        return out


def run(native, d, width, height, K, **kw):
    L = native.lib()
    opts = native.QuantizationOptions(kw.get("dither", False), False, kw.get("color_space", 2), kw.get("kmeans_niter", 0),
                                      kw.get("kmeans_max_samples", 512 ** 2), False)
    pal = np.zeros((K, 3), dtype=np.float64, order="F")
    code = C.c_int(9)
    L.patolette_amd_device(width, height, d.img, d.w, K, C.byref(opts), pal.ctypes.data_as(dp), d.map, 1, C.byref(code))
    assert code.value == 0, native.last_error()
    return pal, d.host_map(), native.last_stats()


def check_map_is_nearest(ob, img_flat, n, pal_srgb, pmap, sample=200000, seed=0):
    """The map must send every pixel to its nearest palette colour in ICtCp (patolette.c:300-324).
    Verified exactly on a random sample with the oracle's conversion + brute-force NN."""
    rng = np.random.default_rng(seed)
    idx = np.sort(rng.choice(n, size=min(sample, n), replace=False))
    sub = np.concatenate([img_flat[idx], img_flat[n + idx], img_flat[2 * n + idx]])
    sub_ict = ob.convert("srgb_to_ictcp", sub)
    k = pal_srgb.shape[0]
    pal_ict = ob.convert("srgb_to_ictcp", np.ascontiguousarray(pal_srgb.T).reshape(-1)).reshape(3, k).T
    want = ob.nn_map(sub_ict, len(idx), pal_ict)
    # the returned sRGB palette went ICtCp -> Rec2020 -> sRGB once (round trip ~1e-12): allow ties within that
    got = pmap[idx].astype(np.int64)
    diff = got != want
    if diff.any():
        px = sub_ict.reshape(3, -1).T[diff]
        d_got = np.sum((px - pal_ict[got[diff]]) ** 2, axis=1)
        d_want = np.sum((px - pal_ict[want[diff]]) ** 2, axis=1)
        assert np.all(np.abs(d_got - d_want) <= 1e-9 * (1e-6 + d_want)), int(diff.sum())
    return int(diff.sum())


@pytest.mark.parametrize("cfg", [("c2", 1920, 1080, 0), ("c3", 4096, 4096, 32)], ids=lambda c: c[0])
def test_full_size_ictcp(gpu, native, ob, cfg):
    name, w, h, niter = cfg
    n, K = w * h, 256
    d = Dev(gpu, n, 3)
    try:
        pal, pmap, st = run(native, d, w, h, K, kmeans_niter=niter)
        pal2, pmap2, _ = run(native, d, w, h, K, kmeans_niter=niter)
        # 1. bit-reproducible run to run (binned sums make float atomics order-independent)
        assert np.array_equal(pal, pal2) and np.array_equal(pmap, pmap2)
        # 2. a full palette of distinct, in-gamut colours; every entry used by uniform noise
        assert st["n_clusters"] == K and np.all(pal >= 0) and np.all(pal <= 1)
        assert len(np.unique(np.round(pal, 12), axis=0)) == K
        assert np.array_equal(np.unique(pmap), np.arange(K))
        # 3. the split loop did the reference's amount of work: D_eff ~ log2(K) (8.07)
        assert 7.0 < st["split_px"] / n < 9.5 and st["n_base_clusters"] == 2
        # 4. the map is the exact nearest-colour assignment
        check_map_is_nearest(ob, d.host_image(), n, pal, pmap)
        # 5. quantisation quality: mean squared ICtCp error of 256 colours on uniform noise
        if niter:
            assert st["kmeans_samples"] == 512 ** 2
    finally:
        d.free()


@pytest.mark.parametrize("cfg", [("c2", 1920, 1080, 0), ("c3", 4096, 4096, 32)], ids=lambda c: c[0])
def test_full_size_matches_oracle(gpu, native, ob, cfg):
    """BASELINE configs[1] and configs[2] AT THEIR OWN SIZE against the CPU oracle (seed 0: the image bench.py's `parity`
    record uses): index map bit for bit, palette within north_star's 1e-5 relative -- 1e-9 asserted (patolette.c:157-343)."""
    import os
    name, w, h, niter = cfg
    n, K = w * h, 256
    d = Dev(gpu, n, 0)
    try:
        pal, pmap, st = run(native, d, w, h, K, kmeans_niter=niter)
        img = d.host_image()
    finally:
        d.free()
    assert np.array_equal(img, ob.image(n, 0))              # the device generator and the oracle's make the same image
    ob.set_threads(os.cpu_count() or 1)                     # faiss / FLANN loops on every core; results independent of it
    try:
        ec, pal_o, map_o = ob.patolette(w, h, img, None, K, dither=False, color_space=2, kmeans_niter=niter, kmeans_max_samples=512 ** 2)
    finally:
        ob.set_threads(1)
    assert ec == 0
    assert np.array_equal(pal == -1.0, pal_o == -1.0)
    rel = np.max(np.abs(pal - pal_o)) / np.max(np.abs(pal_o))
    mism = int(np.count_nonzero(pmap != map_o.astype(np.uint8)))
    print("%s: palette max rel %.3g, map mismatches %d / %d" % (name, rel, mism, n))
    assert rel <= 1e-9                                      # tolerance: north_star asks 1e-5
    assert mism == 0                                        # bit-exact index map


def test_kmeans_lowers_distortion_at_full_size(gpu, native, ob):
    w = h = 2048
    n, K = w * h, 64
    d = Dev(gpu, n, 5)
    try:
        img = d.host_image()
        rng = np.random.default_rng(1)
        idx = rng.choice(n, 100000, replace=False)
        sub = ob.convert("srgb_to_ictcp", np.concatenate([img[idx], img[n + idx], img[2 * n + idx]])).reshape(3, -1).T

        def distortion(pal):
            p = ob.convert("srgb_to_ictcp", np.ascontiguousarray(pal.T).reshape(-1)).reshape(3, K).T
            dd = ((sub[:, None, :] - p[None, :, :]) ** 2).sum(-1)
            return dd.min(1).mean()
        pal0, _, _ = run(native, d, w, h, K, kmeans_niter=0)
        pal1, _, _ = run(native, d, w, h, K, kmeans_niter=16)
        assert distortion(pal1) < distortion(pal0)
    finally:
        d.free()


def test_full_size_cieluv_weighted_map(gpu, native, ob):
    """configs[3] geometry (8192x8192, CIELuv, weights) on the NN-map branch."""
    w = h = 8192
    n, K = w * h, 256
    d = Dev(gpu, n, 7, weighted=True)
    try:
        pal, pmap, st = run(native, d, w, h, K, color_space=1)
        assert st["n_clusters"] == K and np.all(pal >= 0) and np.all(pal <= 1)
        check_map_is_nearest(ob, d.host_image(), n, pal, pmap, sample=100000)
    finally:
        d.free()


def test_dither_full_size_prefix_matches_oracle(gpu, ob):
    """4096x4096 Riemersma chain on the GPU (16.8 M serial steps); the walk is causal, so its first
    300 000 steps must equal the oracle's run truncated after 300 000 steps, bit for bit."""
    w = h = 4096
    n, k, steps = w * h, 256, 300000
    flat = ob.image(n, 13)                                   # used as linear Rec2020 values directly
    pal = ob.image(k, 14).reshape(3, k).T.copy()
    want = ob.dither_prefix(flat, w, h, pal, steps)
    got = np.zeros(n, dtype=np.uintp)
    zp = C.POINTER(C.c_size_t)
    assert gpu.patolette_amd_dither(flat.ctypes.data_as(dp), w, h, np.ascontiguousarray(pal.T).reshape(-1).ctypes.data_as(dp), k,
                                    got.ctypes.data_as(zp)) == 0
    visited = want != 0xFFFF
    assert int(visited.sum()) == steps
    assert np.array_equal(got[visited], want[visited])
    assert got.max() < k


def test_mbd_full_size_bit_exact(gpu, native, ob):
    """The three raster scans of the saliency map at 4096x4096 (64 strips pipelined through progress flags): bit for bit
    the sequential scans of the oracle."""
    from tests.util import scene
    rows = cols = 4096
    img = scene(rows, cols, 7).mean(axis=2).astype(np.float32)
    out = np.zeros_like(img)
    fp = C.POINTER(C.c_float)
    assert native.lib().patolette_amd_mbd(rows, cols, img.ctypes.data_as(fp), 3, out.ctypes.data_as(fp)) == 0
    want = ob.mbd(img, 3)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))


def test_u8_full_size_round_trip(gpu):
    """8-bit adaptor at 4096x4096 with the default saliency weighting: every output consistent with the others."""
    import patolette_amd as p
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, size=(4096, 4096, 3), dtype=np.uint8)
    ok, pal8, pmap, quant, pal, msg = p.quantize_u8(img, 256, dither=False)
    assert ok and pmap.dtype == np.uint8 and pmap.shape == (4096, 4096)
    assert np.array_equal(pal8, np.clip(pal * 255, 0, 255).astype(np.uint8))
    assert np.array_equal(quant, pal8[pmap])
    assert len(np.unique(pmap)) == 256
    # the map is the nearest palette entry in ICtCp: spot-check a sample of pixels against a brute-force search
    from oracle import binding as ob
    idx = rng.integers(0, 4096 * 4096, size=20000)
    px = img.reshape(-1, 3)[idx].astype(np.float64) / 255
    ict = ob.convert("srgb_to_ictcp", ob.planar(px)).reshape(3, -1).T
    pal_ict = ob.convert("srgb_to_ictcp", ob.planar(pal)).reshape(3, -1).T
    d = ((ict[:, None, :] - pal_ict[None, :, :]) ** 2).sum(-1)
    got = pmap.reshape(-1)[idx]
    best = d.min(1)
    assert np.all(d[np.arange(len(idx)), got] <= best * (1 + 1e-9) + 1e-18)


@pytest.mark.parametrize("K", [256, 300])
def test_host_entry_map_equals_device_entry_map(gpu, native, K):
    """The C ABI hands the map back as size_t: the narrow device map (u8 for K <= 256, else u32) crosses PCIe in chunks and is
    widened by host threads.  5 Mpx (three chunks, the last one partial): patolette() must return exactly the map the
    device-resident entry leaves in HBM, and the same palette."""
    import patolette_amd as p
    w, h = 2560, 2048 + 3
    n = w * h
    L = native.lib()
    img = L.patolette_amd_malloc(3 * n * 8)
    dmap = L.patolette_amd_malloc(n * 4)
    try:
        assert img and dmap and L.patolette_amd_fill_image(img, n, 11) == 0
        host = np.empty(3 * n)
        assert L.patolette_amd_memcpy_d2h(host.ctypes.data_as(C.c_void_p), img, host.nbytes) == 0
        opts = native.QuantizationOptions(False, False, 2, 0, 512 ** 2, False)
        pal_d = np.zeros((K, 3), dtype=np.float64, order="F")
        code = C.c_int(9)
        me = 1 if K <= 256 else 4
        L.patolette_amd_device(w, h, img, None, K, C.byref(opts), pal_d.ctypes.data_as(dp), dmap, me, C.byref(code))
        assert code.value == 0, native.last_error()
        want = np.empty(n, dtype=np.uint8 if me == 1 else np.uint32)
        assert L.patolette_amd_memcpy_d2h(want.ctypes.data_as(C.c_void_p), dmap, want.nbytes) == 0
        ok, pal_h, map_h, _ = p.quantize(w, h, np.asfortranarray(host.reshape(3, n).T), K, dither=False, tile_size=0, kmeans_niter=0)
        assert ok and map_h.dtype == np.uintp
        assert np.array_equal(pal_h, pal_d)
        assert np.array_equal(map_h, want.astype(np.uintp))
    finally:
        L.patolette_amd_free(img)
        L.patolette_amd_free(dmap)


def test_c4_full_size_cieluv_weighted_dither(gpu, native, ob):
    """BASELINE configs[3] as stated: 8192x8192, 256 colours, CIELuv + weights + Riemersma dither, through the
    device-resident entry.  The chain is causal, so the first 300 000 steps of the map must equal, bit for bit, the
    oracle's chain run on the inputs the device stage saw (the device's own Rec2020 pixels -- the same two conversion
    kernels applied to a host copy -- and the palette as the mapping stage used it); over the whole image every index
    is a palette row, every row is used, and error diffusion keeps the mean colour."""
    w = h = 8192
    n, K, steps = w * h, 256, 300000
    d = Dev(gpu, n, 7, weighted=True)
    try:
        pal, pmap, st = run(native, d, w, h, K, color_space=1, dither=True)
        L = native.lib()
        assert st["n_clusters"] == K and np.all(pal >= 0) and np.all(pal <= 1)
        mp = np.zeros((K, 3), dtype=np.float64, order="F")
        assert L.patolette_amd_last_map_palette(mp.ctypes.data_as(dp), K) == K
        rec = d.host_image()
        assert L.patolette_amd_convert(1, rec.ctypes.data_as(dp), n) == 0      # sRGB -> CIELuv (patolette.c:201-207)
        assert L.patolette_amd_convert(3, rec.ctypes.data_as(dp), n) == 0      # CIELuv -> linear Rec2020 (patolette.c:276-281)
        want = ob.dither_prefix(rec, w, h, mp, steps)
        visited = want != 0xFFFF
        assert int(visited.sum()) == steps
        assert np.array_equal(pmap[visited].astype(np.uintp), want[visited])
        assert pmap.max() < K and len(np.unique(pmap)) == K
        # error diffusion: the dithered image keeps the mean colour of the original (in the space it is diffused in)
        mean_img = rec.reshape(3, n).mean(axis=1)
        counts = np.bincount(pmap, minlength=K).astype(np.float64)
        mean_map = (counts[:, None] * mp).sum(axis=0) / n
        assert np.max(np.abs(mean_img - mean_map)) < 2e-3, (mean_img, mean_map)
    finally:
        d.free()


@pytest.mark.parametrize("dither", [True, False], ids=["c4", "c4map"])
def test_c4_full_size_matches_oracle(gpu, native, ob, dither):
    """BASELINE configs[3] AT ITS OWN SIZE against the CPU oracle: 8192 x 8192, 256 colours, CIELuv + weights, with the
    Riemersma dither (every one of the 67 108 864 chain steps; riemersma.c:259-341) and on the NN-map branch
    (patolette.c:300-324).  Seed 0: the image and weights `bench.py --config c4 / c4map` takes its `parity` record on.
    Index map bit for bit, palette within north_star's 1e-5 relative -- 1e-9 asserted."""
    import os
    w = h = 8192
    n, K = w * h, 256
    d = Dev(gpu, n, 0, weighted=True)
    try:
        pal, pmap, st = run(native, d, w, h, K, color_space=1, dither=dither)
        img = d.host_image()
        wts = np.empty(n)
        assert gpu.patolette_amd_memcpy_d2h(wts.ctypes.data_as(C.c_void_p), d.w, wts.nbytes) == 0
    finally:
        d.free()
    assert np.array_equal(wts, ob.weights(n, 0))
    ob.set_threads(os.cpu_count() or 1)
    try:
        ec, pal_o, map_o = ob.patolette(w, h, img, wts, K, dither=dither, color_space=1, kmeans_niter=0)
    finally:
        ob.set_threads(1)
    assert ec == 0
    assert np.array_equal(pal == -1.0, pal_o == -1.0)
    rel = np.max(np.abs(pal - pal_o)) / np.max(np.abs(pal_o))
    mism = int(np.count_nonzero(pmap != map_o.astype(np.uint8)))
    print("c4 dither=%s: palette max rel %.3g, map mismatches %d / %d, oracle stages %s, GPU ms_map %.2f, runs %d repairs %d"
          % (dither, rel, mism, n, ob.last_timings(), st["ms_map"], st["dither_segments"], st["dither_repairs"]))
    assert rel <= 1e-9
    assert mism == 0
    if dither:
        assert st["dither_segments"] > 1000                                   # (the stage's time is bench.py's business: profiles/*_bench_c4.json)
