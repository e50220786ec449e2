"""Randomised sweep of the dither stage alone (GPU box): images of 65 536 .. 400 000 pixels of varied shape and content, palettes
of 8 .. 256 rows out of the oracle's quantiser, random numbers of runs and warm-up lengths, both layouts (one lane / one
wavefront per run) -- every map against the oracle's serial chain.  usage: fuzz_dither.py [seed] [cases]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import binding as ob  # noqa: E402
from patolette_amd import _native  # noqa: E402
from tests.util import scene  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
L = _native.lib()
dp, zp = C.POINTER(C.c_double), C.POINTER(C.c_size_t)
bad = 0
for case in range(ncases):
    n_target = int(rng.integers(65536, 400000))
    w = int(rng.integers(16, 2000))
    h = max(1, n_target // w)
    if w * h < 65536:
        h = 65536 // w + 1
    n = w * h
    kind = rng.choice(["noise", "scene", "post", "flat", "gradient", "nearflat"])
    if kind == "noise":
        img = rng.random((h, w, 3))
    elif kind in ("scene", "post"):
        img = scene(h, w, int(rng.integers(0, 1000)))
        if kind == "post":
            img = np.round(img * 5.0) / 5.0
    elif kind == "flat":
        img = np.tile(rng.random(3), (h, w, 1))
    elif kind == "gradient":
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        img = np.stack([xx / w, yy / h, (xx + yy) / (w + h)], axis=2)
    else:
        img = np.clip(rng.random(3) + 0.01 * (rng.random((h, w, 3)) - 0.5), 0, 1)
    srgb = np.concatenate([img[:, :, c].reshape(-1) for c in range(3)])
    K = int(rng.choice([8, 9, 16, 31, 64, 100, 200, 256]))
    ec, pal, _ = ob.patolette(w, h, srgb, None, K, dither=False, color_space=int(rng.integers(0, 3)), kmeans_niter=0)
    assert ec == 0
    pal = pal[pal[:, 0] >= 0]
    if pal.shape[0] < 8:
        pal = np.vstack([pal, rng.random((8 - pal.shape[0], 3))])
    pal = ob.convert("srgb_to_rec2020", ob.planar(pal).copy()).reshape(3, -1).T.copy()
    flat = ob.convert("srgb_to_rec2020", srgb)
    want = ob.dither(flat, w, h, pal)
    p = np.ascontiguousarray(pal.T).reshape(-1)
    for layout in (1, 0):
        S = int(rng.choice([0, 0, 2, 7, 100, 1000, int(rng.integers(2, 6000))]))
        W = int(rng.choice([-1, -1, 0, 16, 64, int(rng.integers(0, 700))]))
        L.patolette_amd_dither_layout(layout)
        L.patolette_amd_dither_config(S, W)
        got = np.zeros(n, dtype=np.uintp)
        assert L.patolette_amd_dither(flat.ctypes.data_as(dp), w, h, p.ctypes.data_as(dp), pal.shape[0], got.ctypes.data_as(zp)) == 0, _native.last_error()
        st = _native.last_stats()
        mism = int(np.sum(got != want))
        line = "case %3d %4dx%-4d %-8s K=%3d layout=%d S=%5d W=%4d -> runs %6d repairs %5d passes %d  mismatches %d" % (
            case, w, h, kind, pal.shape[0], layout, S, W, st["dither_segments"], st["dither_repairs"], st["dither_rounds"], mism)
        if mism:
            bad += 1
            print("BAD ", line, flush=True)
        elif case % 5 == 0:
            print("ok  ", line, flush=True)
L.patolette_amd_dither_config(0, -1)
L.patolette_amd_dither_layout(-1)
print("%d cases x 2 layouts, %d with differences" % (ncases, bad))
