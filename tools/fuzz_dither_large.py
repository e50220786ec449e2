"""Randomised sweep of the dither stage at the sizes where the lane-per-run layout is the DEFAULT (>= 2^23 pixels; GPU box):
8.4 .. 12 Mpx images of varied shape and content -- noise, a scene, a posterised scene, gradients, nearly flat, flat stretches
whose colour is not a palette entry --, palettes of 8 .. 256 rows taken from a reduced copy of the image by the oracle's quantiser,
default knobs: every map entry against the oracle's serial chain.  usage: fuzz_dither_large.py [seed] [cases]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import binding as ob  # noqa: E402
from patolette_amd import _native  # noqa: E402
from tests.util import scene  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 6
L = _native.lib()
dp, zp = C.POINTER(C.c_double), C.POINTER(C.c_size_t)
kinds = ["noise", "scene", "post", "gradient", "nearflat", "halfflat", "bands"]
bad = 0
for case in range(ncases):
    n_target = int(rng.integers(1 << 23, 12_000_000))
    w = int(rng.integers(1500, 6000))
    h = n_target // w + 1
    n = w * h
    kind = kinds[case % len(kinds)]
    if kind == "noise":
        img = rng.random((h, w, 3))
    elif kind in ("scene", "post"):
        img = scene(h, w, int(rng.integers(0, 1000)))
        if kind == "post":
            img = np.round(img * 6.0) / 6.0
    elif kind == "gradient":
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        img = np.stack([xx / w, yy / h, (xx + yy) / (w + h)], axis=2)
    elif kind == "nearflat":
        img = np.clip(rng.random(3) + 0.01 * (rng.random((h, w, 3)) - 0.5), 0, 1)
    elif kind == "halfflat":                                # one half a scene, the other one colour (not in the palette)
        img = scene(h, w, int(rng.integers(0, 1000)))
        img[:, w // 2:] = rng.random(3)
    else:                                                   # bands of flat colour, 200 rows each
        img = np.zeros((h, w, 3))
        for y0 in range(0, h, 200):
            img[y0:y0 + 200] = rng.random(3)
    K = int(rng.choice([8, 16, 64, 200, 256]))
    small = img[::8, ::8]
    sh, sw = small.shape[:2]
    ssrgb = np.concatenate([small[:, :, c].reshape(-1) for c in range(3)])
    ec, pal, _ = ob.patolette(sw, sh, ssrgb, None, K, dither=False, color_space=int(rng.integers(0, 3)), kmeans_niter=0)
    assert ec == 0
    pal = pal[pal[:, 0] >= 0]
    if pal.shape[0] < 8:
        pal = np.vstack([pal, rng.random((8 - pal.shape[0], 3))])
    pal = ob.convert("srgb_to_rec2020", ob.planar(pal).copy()).reshape(3, -1).T.copy()
    srgb = np.concatenate([img[:, :, c].reshape(-1) for c in range(3)])
    del img
    flat = ob.convert("srgb_to_rec2020", srgb)
    del srgb
    t0 = time.time()
    want = ob.dither(flat, w, h, pal)
    t_o = time.time() - t0
    p = np.ascontiguousarray(pal.T).reshape(-1)
    got = np.zeros(n, dtype=np.uintp)
    t0 = time.time()
    assert L.patolette_amd_dither(flat.ctypes.data_as(dp), w, h, p.ctypes.data_as(dp), pal.shape[0], got.ctypes.data_as(zp)) == 0, _native.last_error()
    t_g = time.time() - t0
    st = _native.last_stats()
    mism = int(np.sum(got != want))
    bad += mism != 0
    print("%s case %2d %4dx%-4d %-8s K=%3d -> runs %6d repairs %5d passes %2d through %d  mismatches %d  (oracle %.1f s, entry %.2f s host to host)" % (
        "BAD " if mism else "ok  ", case, w, h, kind, pal.shape[0], st["dither_segments"], st["dither_repairs"], st["dither_rounds"],
        st["dither_through"], mism, t_o, t_g), flush=True)
print("%d cases, %d with differences" % (ncases, bad))
