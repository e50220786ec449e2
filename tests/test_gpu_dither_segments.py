"""Segment-parallel Riemersma dither (map.hip, DitherSeg) against the oracle's serial chain
(oracle/patolette_oracle.c orc_dither_riemersma <- lib/src/dither/riemersma.c:259-341,360-373,437-459).

The curve is cut into runs walked side by side from speculative warm-ups; every boundary is verified and a run whose
starting state was not the chain's is walked again from the state rebuilt out of the map.  Whatever the number of runs
and however short the warm-up, the map must be the reference's chain bit for bit.
"""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

dp = C.POINTER(C.c_double)
zp = C.POINTER(C.c_size_t)


def _d(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(params=[1, 0], ids=["lane-per-run", "wavefront-per-run"])
def cfg(gpu, request):
    """Sets the dither knob for one test and always puts the defaults back.  Every test runs under both layouts: one lane per
    run wherever it applies (8 <= K <= 256, >= 65 536 pixels; the default) and one wavefront per run everywhere."""
    gpu.patolette_amd_dither_layout(request.param)

    def set_(segments, warm=-1):
        gpu.patolette_amd_dither_config(int(segments), int(warm))
    yield set_
    gpu.patolette_amd_dither_config(0, -1)
    gpu.patolette_amd_dither_layout(-1)


def _dither(gpu, native, flat, w, h, pal):
    k = pal.shape[0]
    got = np.zeros(w * h, dtype=np.uintp)
    p = np.ascontiguousarray(pal.T).reshape(-1)
    assert gpu.patolette_amd_dither(_d(flat), w, h, _d(p), k, got.ctypes.data_as(zp)) == 0, native.last_error()
    return got, native.last_stats()


def _noise_case(ob, w, h, k, seed=7):
    n = w * h
    flat = ob.convert("srgb_to_rec2020", ob.image(n, seed))
    pal = ob.convert("srgb_to_rec2020", ob.image(k, seed + 2)).reshape(3, k).T.copy()
    return flat, pal


@pytest.mark.parametrize("wh", [(256, 256), (300, 420), (130, 70), (1000, 100), (17, 4000)])
@pytest.mark.parametrize("k", [16, 100, 256, 700])
def test_every_number_of_runs_gives_the_serial_chain(gpu, native, ob, cfg, wh, k):
    w, h = wh
    flat, pal = _noise_case(ob, w, h, k)
    want = ob.dither(flat, w, h, pal)
    for seg in (1, 2, 7, 1024, 0):
        cfg(seg)
        got, st = _dither(gpu, native, flat, w, h, pal)
        assert np.array_equal(got, want), "%dx%d k=%d S=%d: %d mismatches, stats %s" % (w, h, k, seg, int(np.sum(got != want)), st)
        if seg in (1, 2, 7):
            assert st["dither_segments"] == seg
        if seg == 1024:
            assert 1 < st["dither_segments"] <= 1024
        if st["dither_segments"] > 1:
            assert st["dither_rounds"] >= 1


@pytest.mark.parametrize("warm", [0, 5, 16, 40, 100])
def test_a_missed_speculation_is_repaired_to_the_same_map(gpu, native, ob, cfg, warm):
    """warm = 0: every run starts from a zero queue where the chain's is not zero -> every boundary fails its check and is
    rebuilt from the map; short warm-ups (they start at the aligned 64-position block that holds the pixel asked for, so
    `warm` .. `warm` + 63 steps) fail some.  Same map either way."""
    w, h, k = 420, 300, 256
    flat, pal = _noise_case(ob, w, h, k, seed=11)
    want = ob.dither(flat, w, h, pal)
    for seg in (2, 7, 64, 900):
        cfg(seg, warm)
        got, st = _dither(gpu, native, flat, w, h, pal)
        assert np.array_equal(got, want), "warm %d S=%d: %d mismatches, stats %s" % (warm, seg, int(np.sum(got != want)), st)
        S = st["dither_segments"]
        if warm == 0:
            assert st["dither_repairs"] >= S - 1 and st["dither_rounds"] >= 2, st
    cfg(0, -1)
    got, st = _dither(gpu, native, flat, w, h, pal)
    assert np.array_equal(got, want)
    print("default knob on %dx%d: %s" % (w, h, {k_: st[k_] for k_ in ("dither_segments", "dither_repairs", "dither_rounds")}))


@pytest.mark.parametrize("content", ["scene", "posterised", "nearflat", "flat", "gradient"])
@pytest.mark.parametrize("k", [12, 256])
def test_content_classes(gpu, native, ob, cfg, content, k):
    """Where the chain's errors are small or periodic the speculative chains take longest to meet the true one."""
    rows, cols = 384, 512
    n = rows * cols
    sc = util.scene(rows, cols, 3)
    if content == "posterised":
        sc = np.round(sc * 7.0) / 7.0
    elif content == "nearflat":
        sc = np.clip(0.5 + 0.01 * (ob.image(n, 9).reshape(3, n).T.reshape(rows, cols, 3) - 0.5), 0, 1)
    elif content == "flat":
        sc = np.full((rows, cols, 3), 0.3)
    elif content == "gradient":
        yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
        sc = np.stack([xx / cols, yy / rows, (xx + yy) / (rows + cols)], axis=2)
    srgb = np.concatenate([sc[:, :, c].reshape(-1) for c in range(3)])
    ec, pal, _ = ob.patolette(cols, rows, srgb, None, k, dither=False, color_space=2, kmeans_niter=0)
    assert ec == 0
    pal = pal[pal[:, 0] >= 0]
    pal = ob.convert("srgb_to_rec2020", ob.planar(pal).copy()).reshape(3, -1).T.copy()
    flat = ob.convert("srgb_to_rec2020", srgb)
    want = ob.dither(flat, cols, rows, pal)
    for seg, warm in ((0, -1), (97, 64), (1024, 256)):
        cfg(seg, warm)
        got, st = _dither(gpu, native, flat, cols, rows, pal)
        assert np.array_equal(got, want), "%s k=%d S=%d warm=%d: %d mismatches, %s" % (content, k, seg, warm, int(np.sum(got != want)), st)


def test_four_megapixels_default_knob(gpu, native, ob, cfg):
    """2048 x 2048, 256 colours, every pixel against the oracle's chain; the default cut (one run per ~1000 pixels up to eight
    per compute unit) with the default warm-up."""
    w = h = 2048
    flat, pal = _noise_case(ob, w, h, 256, seed=23)
    want = ob.dither(flat, w, h, pal)
    cfg(0)
    got, st = _dither(gpu, native, flat, w, h, pal)
    assert np.array_equal(got, want), (int(np.sum(got != want)), st)
    assert st["dither_segments"] > 1000
    print("2048^2: %s" % {k_: st[k_] for k_ in ("dither_segments", "dither_repairs", "dither_rounds")})


@pytest.mark.parametrize("cs,weighted", [(1, True), (2, False), (0, False)])
def test_full_path_with_dither_on_through_the_c_abi(gpu, native, ob, cs, weighted):
    """patolette() with dither = true (the reference's default, patolette.c:112): palette and every map entry."""
    w, h, K = 640, 480, 256
    n = w * h
    flat = ob.image(n, 31)
    wt = ob.weights(n, 31) if weighted else None
    L = native.lib()
    opts = native.QuantizationOptions(True, False, cs, 0, 512 ** 2, False)
    pal = np.zeros((K, 3), dtype=np.float64, order="F")
    pmap = np.zeros(n, dtype=np.uintp)
    code = C.c_int(9)
    L.patolette(w, h, _d(flat), _d(wt) if wt is not None else None, K, C.byref(opts), _d(pal), pmap.ctypes.data_as(zp), C.byref(code))
    assert code.value == 0, native.last_error()
    st = native.last_stats()
    ec, pal_o, map_o = ob.patolette(w, h, flat, wt, K, dither=True, color_space=cs, kmeans_niter=0)
    assert ec == 0
    assert np.max(np.abs(pal - pal_o)) <= 1e-9 * max(1.0, np.max(np.abs(pal_o)))
    assert np.array_equal(pmap, map_o), (int(np.sum(pmap != map_o)), st)
    assert st["dither_segments"] > 1


@pytest.mark.parametrize("shape", ["flat", "half", "bands"])
@pytest.mark.parametrize("k", [4, 16])
def test_flat_stretches_off_the_palette(gpu, native, ob, cfg, shape, k):
    """Where the image is flat over more than a warm-up and the flat colour is NOT a palette entry, the true chain is periodic
    and a zero-queue chain settles into the same cycle at another phase: it never meets the true one (measured with the
    oracle: period 4 .. 14, one start in ten in phase).  Verification passes then fix one run each; the product notices the
    stall and walks the lowest unverified run through its successors (one wavefront, the true chain) until it meets what is
    there.  Same map, and far fewer passes than runs."""
    w, h = 512, 300
    n = w * h
    rng = np.random.default_rng(40 + k)
    pal = rng.random((k if k >= 8 else 8, 3))                       # (8 rows at least: the lane layout's lower bound)
    pal[k:] = 5.0 + rng.random((pal.shape[0] - k, 3))               # ... the extra ones far away: never chosen
    img = np.tile(rng.random(3), (h, w, 1))
    if shape == "half":
        img[:, w // 2:] = rng.random((h, w - w // 2, 3))
    elif shape == "bands":
        for y0 in range(0, h, 60):
            img[y0:y0 + 60] = rng.random(3)
    flat = np.concatenate([img[:, :, c].reshape(-1) for c in range(3)])
    want = ob.dither(flat, w, h, pal)
    for seg in (0, 300):
        cfg(seg)
        got, st = _dither(gpu, native, flat, w, h, pal)
        assert np.array_equal(got, want), "%s k=%d S=%d: %d mismatches, %s" % (shape, k, seg, int(np.sum(got != want)), st)
        assert st["dither_rounds"] <= 40, st                        # (not one pass per run)
        print("%s k=%d S=%d: %s" % (shape, k, seg, {q: st[q] for q in ("dither_segments", "dither_repairs", "dither_rounds", "dither_through")}))
