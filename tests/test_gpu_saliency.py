"""Saliency-derived weights: the HIP stage against the CPU restatement of the
reference's `get_weights` (oracle/saliency.py + orc_mbd).

Bars: the minimum-barrier map is f32 min/max/sub only -> bit-exact; the weights are f64 through
pow/cbrt/exp and reductions in another order -> 1e-9 relative (the oracle itself is numpy, whose
summation order the device does not reproduce).
"""
import ctypes as C

import numpy as np
import pytest

from tests.util import scene

pytestmark = pytest.mark.gpu
fp = C.POINTER(C.c_float)


def _mbd_gpu(native, img32, iters=3):
    img32 = np.ascontiguousarray(img32, dtype=np.float32)
    rows, cols = img32.shape
    out = np.zeros((rows, cols), dtype=np.float32)
    rc = native.lib().patolette_amd_mbd(rows, cols, img32.ctypes.data_as(fp), iters, out.ctypes.data_as(fp))
    return rc, out


@pytest.mark.parametrize("rows,cols", [(4, 4), (5, 9), (64, 64), (66, 67), (67, 130), (129, 70), (200, 333), (131, 1031)])
@pytest.mark.parametrize("kind", ["noise", "scene"])
def test_mbd_bit_exact(gpu, native, ob, rows, cols, kind):
    rng = np.random.default_rng(rows * 1000 + cols)
    if kind == "noise":
        img = rng.random((rows, cols), dtype=np.float32)
    else:
        img = scene(rows, cols, 3).mean(axis=2).astype(np.float32)
    for iters in (1, 2, 3):
        rc, got = _mbd_gpu(native, img, iters)
        want = ob.mbd(img, iters)
        assert rc == 0 and np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rows, cols, iters)


def test_mbd_small_shapes_rejected(gpu, native):
    rc, _ = _mbd_gpu(native, np.zeros((3, 50), dtype=np.float32))
    assert rc == -2


@pytest.mark.parametrize("rows,cols,tile", [(61, 83, 512.0), (120, 200, 64.0), (257, 190, 128.0)])
def test_saliency_weights_match_oracle(gpu, ob, rows, cols, tile):
    import patolette_amd as p
    from oracle import saliency
    img = scene(rows, cols, 11)
    want = saliency.get_weights(img, tile)
    got = p.saliency_weights(cols, rows, img.reshape(-1, 3), tile)
    assert got.shape == (rows * cols,) and np.all(got >= 1.0)
    assert np.allclose(got, want, rtol=1e-9, atol=0), np.max(np.abs(got - want) / want)


def test_saliency_error_paths(gpu):
    import patolette_amd as p
    rng = np.random.default_rng(1)
    with pytest.raises(ValueError):
        p.saliency_weights(9, 9, rng.random((81, 3)))                    # fewer than 100 pixels: empty border band
    with pytest.raises(ValueError):
        p.saliency_weights(3, 400, rng.random((1200, 3)))                # a side <= 3
    with pytest.raises(ValueError):
        p.saliency_weights(2000, 6, rng.random((12000, 3)))              # band of 10 rows does not fit 6 rows
    flat = np.full((40 * 40, 3), 0.25)
    with pytest.raises(np.linalg.LinAlgError):
        p.saliency_weights(40, 40, flat)                                 # constant border: singular covariance
    with pytest.raises(np.linalg.LinAlgError):
        p.quantize(40, 40, flat, 8)


@pytest.mark.parametrize("cs,dither", [(2, False), (1, True)])
def test_quantize_default_tile_size_end_to_end(gpu, ob, cs, dither):
    """quantize(tile_size=512) == the reference's flow: get_weights, then patolette() with those weights."""
    import patolette_amd as p
    from oracle import saliency
    rows, cols, K = 150, 210, 24
    img = scene(rows, cols, 5)
    colors = img.reshape(-1, 3)
    ok, pal, pmap, msg = p.quantize(cols, rows, colors, K, dither=dither, color_space=cs, tile_size=96, kmeans_niter=4,
                                    kmeans_max_samples=8192)
    assert ok
    # same pipeline fed with the device-derived weights explicitly: identical
    w_gpu = p.saliency_weights(cols, rows, colors, 96)
    ok2, pal2, pmap2, _ = p.quantize(cols, rows, colors, K, dither=dither, color_space=cs, tile_size=0, kmeans_niter=4,
                                     kmeans_max_samples=8192, weights=w_gpu)
    assert np.array_equal(pal, pal2) and np.array_equal(pmap, pmap2)
    # the oracle's flow end to end
    w_cpu = saliency.get_weights(img, 96.0)
    ec, pal_o, pmap_o = ob.patolette(cols, rows, ob.planar(colors), w_cpu, K, dither=dither, color_space=cs, kmeans_niter=4,
                                     kmeans_max_samples=8192)
    assert ec == 0
    assert np.allclose(pal, pal_o, rtol=0, atol=1e-6)
    assert np.mean(pmap == pmap_o) > 0.999
    # u8 entry with the default-style tile size
    img8 = np.round(img * 255).astype(np.uint8)
    ok3, pal8, pm8, q8, palf, _ = p.quantize_u8(img8, K, dither=dither, color_space=cs, tile_size=96, kmeans_niter=4,
                                                kmeans_max_samples=8192)
    c8 = img8.reshape(-1, 3).astype(np.float64)
    c8 /= 255
    ok4, pal4, pm4, _ = p.quantize(cols, rows, c8, K, dither=dither, color_space=cs, tile_size=96, kmeans_niter=4,
                                   kmeans_max_samples=8192)
    assert ok3 and ok4 and np.array_equal(palf, pal4) and np.array_equal(pm8.reshape(-1).astype(np.uintp), pm4)
