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
    run wherever it applies (8 <= K <= 256; from 65 536 pixels on under this knob, from 2^23 by default) and one wavefront per run
    everywhere."""
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
def test_full_path_with_dither_on_through_the_c_abi(gpu, native, ob, cfg, cs, weighted):
    """patolette() with dither = true (the reference's default, patolette.c:112): palette and every map entry -- under both layouts
    (`cfg`): with one lane per run the conversion of the pixels to linear Rec2020 (patolette.c:274-287) rides on the gather into
    curve order (k_dither_streams<sRGB | CIELuv | ICtCp>), with one wavefront per run it is a pass of its own."""
    cfg(0)
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
        print("%s k=%d S=%d: %s" % (shape, k, seg, {q: st[q] for q in ("dither_segments", "dither_repairs", "dither_rounds", "dither_through",
                                                                       "dither_jumps", "dither_solo")}))
        # the machinery that makes this fast must have RUN (a change that silently disables it keeps the map right, ten times slower):
        lanes = gpu.patolette_amd_dither_layout_in_use(w, h, pal.shape[0]) == 1
        if lanes and st["dither_repairs"] > 0:
            assert st["dither_jumps"] >= 1, st                      # a walk over one colour found its period and wrote the pattern
        if not lanes and st["dither_rounds"] >= 5:
            assert st["dither_through"] >= 1, st                    # the stalled passes were answered by a walk through the successors


def test_bands_with_many_failing_boundaries(gpu, native, ob, cfg):
    """Twelve flat bands, twelve palette rows, none of them a band's colour: every band is a periodic stretch with thousands of failing
    boundaries.  Lane layout: the rows' wavefronts find the period and write the pattern (`dither_jumps`), a handful of passes;
    wavefront layout: walks through the successors (`dither_through`).  Same map as the serial chain."""
    w, h, k = 2048, 1536, 12
    rng = np.random.default_rng(12)
    pal = rng.random((k, 3))
    img = np.zeros((h, w, 3))
    for y0 in range(0, h, h // 12):
        img[y0:y0 + h // 12] = rng.random(3)
    flat = np.concatenate([img[:, :, c].reshape(-1) for c in range(3)])
    want = ob.dither(flat, w, h, pal)
    cfg(0)
    got, st = _dither(gpu, native, flat, w, h, pal)
    print("twelve bands: %s" % {q: st[q] for q in ("dither_segments", "dither_repairs", "dither_rounds", "dither_through", "dither_jumps", "dither_solo")})
    assert np.array_equal(got, want), (int(np.sum(got != want)), st)
    if gpu.patolette_amd_dither_layout_in_use(w, h, k) == 1:
        assert st["dither_jumps"] >= 12 and st["dither_rounds"] <= 8, st
    else:
        assert st["dither_through"] >= 1, st


@pytest.mark.parametrize("shape", ["bands", "noise+flat"])
def test_one_wavefront_alone_gives_the_serial_chain(gpu, native, ob, shape):
    """The lane layout's answer to passes that stop making progress: ONE wavefront from the lowest failing boundary through everything
    in its way (k_dither_lane_repair, solo).  With the jumps in place no image found so far stalls the passes (tools/diag/solo_probe.py),
    so the walk is forced here (patolette_amd_debug_dither_stall_passes(0): every repair pass is such a walk) and held to the oracle."""
    w, h, k = 640, 400, 12
    rng = np.random.default_rng(21)
    pal = rng.random((k, 3))
    img = rng.random((h, w, 3))
    if shape == "bands":
        for y0 in range(0, h, 40):
            img[y0:y0 + 40] = rng.random(3)
    else:
        img[h // 3:2 * h // 3] = rng.random(3)
    flat = np.concatenate([img[:, :, c].reshape(-1) for c in range(3)])
    want = ob.dither(flat, w, h, pal)
    gpu.patolette_amd_dither_layout(1)
    gpu.patolette_amd_dither_config(0, 0)                            # no warm-up: every boundary fails the first check
    prev = gpu.patolette_amd_debug_dither_stall_passes(0)
    try:
        got, st = _dither(gpu, native, flat, w, h, pal)
    finally:
        gpu.patolette_amd_debug_dither_stall_passes(prev)
        gpu.patolette_amd_dither_config(0, -1)
        gpu.patolette_amd_dither_layout(-1)
    print("solo %s: %s" % (shape, {q: st[q] for q in ("dither_segments", "dither_repairs", "dither_rounds", "dither_through", "dither_jumps", "dither_solo")}))
    assert np.array_equal(got, want), (int(np.sum(got != want)), st)
    assert st["dither_solo"] >= 1, st


@pytest.mark.parametrize("k", [8, 64])
def test_adversarial_checkerboard_of_flat_tiles(gpu, native, ob, cfg, k):
    """Tiles of one colour shorter than a run, none of the colours a palette entry (2048^2, tile sides 8 .. 64): every run starts inside
    some flat stretch, the speculative chains settle into the stretch's cycle at the wrong phase and almost every boundary fails the
    first check.  The map must be the serial chain's bit for bit, and the mapping stage must stay within 25x of what noise costs."""
    w = h = 2048
    n = w * h
    rng = np.random.default_rng(300 + k)
    pal = rng.random((k, 3))
    img = np.zeros((h, w, 3))
    y = 0
    while y < h:
        side = int(rng.integers(8, 65))
        x = 0
        while x < w:
            sx = int(rng.integers(8, 65))
            img[y:y + side, x:x + sx] = rng.random(3)
            x += sx
        y += side
    flat = np.concatenate([img[:, :, c].reshape(-1) for c in range(3)])
    want = ob.dither(flat, w, h, pal)
    cfg(0)
    noise, _ = _noise_case(ob, w, h, k, seed=9)
    import time
    _dither(gpu, native, noise, w, h, pal)                          # (warm-up: workspace, curve order)
    t0 = time.perf_counter()
    _dither(gpu, native, noise, w, h, pal)
    t_noise = time.perf_counter() - t0
    t0 = time.perf_counter()
    got, st = _dither(gpu, native, flat, w, h, pal)
    t_adv = time.perf_counter() - t0
    print("checkerboard k=%d: %.1f ms against %.1f ms for noise; %s" % (k, 1e3 * t_adv, 1e3 * t_noise, {q: st[q] for q in (
        "dither_segments", "dither_repairs", "dither_rounds", "dither_through", "dither_jumps", "dither_solo")}))
    assert np.array_equal(got, want), (int(np.sum(got != want)), st)
    assert t_adv <= 25.0 * t_noise, (t_adv, t_noise, st)


def test_lane_layout_gives_up_and_the_wavefront_layout_takes_over(gpu, native, ob):
    """patolette_amd_debug_dither_solo_cap(0): the lane layout returns at its first stalled pass and the wavefront layout starts the
    image over (launch_dither's fall-back, with the pixels converted by k_dither_convert) -- through the full path, CIELuv, so that the
    fall-back's own conversion is what the oracle is compared with."""
    import patolette_amd as p
    w, h, K = 1024, 512, 8
    rng = np.random.default_rng(77)
    img = np.tile(rng.random(3), (h, w, 1))
    img[:, 3 * w // 4:] = rng.random((h, w - 3 * w // 4, 3))       # three quarters one flat colour: more colours than palette rows
    colors = img.reshape(-1, 3)
    gpu.patolette_amd_dither_layout(1)
    prev = gpu.patolette_amd_debug_dither_solo_cap(0)
    try:
        ok, pal, pmap, msg = p.quantize(w, h, colors, K, dither=True, color_space=p.ColorSpace_CIELuv, tile_size=0, kmeans_niter=0)
        st = p.last_stats()
    finally:
        gpu.patolette_amd_debug_dither_solo_cap(prev)
        gpu.patolette_amd_dither_layout(-1)
    assert ok, msg
    print("forced fall-back: %s" % {q: st[q] for q in ("dither_segments", "dither_repairs", "dither_rounds", "dither_through", "dither_solo")})
    ec, pal_o, map_o = ob.patolette(w, h, ob.planar(colors), None, K, dither=True, color_space=1, kmeans_niter=0)
    assert ec == 0 and np.allclose(pal, pal_o, rtol=0, atol=1e-9)
    assert np.array_equal(pmap, map_o), int(np.sum(pmap != map_o))
