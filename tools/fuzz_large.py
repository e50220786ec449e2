"""Randomised parity sweep of the LARGE-image paths (GPU box): the LDS-table NN map (>= 4 Mpx) and the LDS-table KMeans
assignment (forced from 2 M samples) on palettes / data of varied structure, against the oracle's full scans (threaded).
  python tools/fuzz_large.py [seed] [cases]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
os.environ.setdefault("PAMD_KM_G64_MIN", "1")           # the 64^3 grid + LDS table from 2 M samples on (normally 8 M)
from patolette_amd import _native  # noqa: E402
from oracle import binding as ob  # noqa: E402

L = _native.lib()
dp, zp = C.POINTER(C.c_double), C.POINTER(C.c_size_t)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ob.set_threads(os.cpu_count() or 1)


def d(a):
    return a.ctypes.data_as(dp)


def pixels(kind, n):
    if kind == "ictcp":
        return ob.convert("srgb_to_ictcp", ob.image(n, int(rng.integers(1, 1000))))
    if kind == "luv":
        return ob.convert("srgb_to_cieluv", ob.image(n, int(rng.integers(1, 1000))))
    if kind == "blobs":                                  # a few tight colour clusters + background noise
        k = int(rng.integers(3, 12))
        cen = rng.random((k, 3))
        a = rng.integers(0, k, size=n)
        pts = cen[a] + rng.standard_normal((n, 3)) * rng.uniform(0.002, 0.05, size=(k, 1))[a]
        return np.ascontiguousarray(pts.T).reshape(-1)
    if kind == "dark":                                   # near the origin: no cancellation in the f32 distance form
        return np.ascontiguousarray((rng.random((n, 3)) ** 3 * 0.05).T).reshape(-1)
    if kind == "quantised":                              # 8-bit-like lattice: many coincident points and exact ties
        return np.ascontiguousarray((rng.integers(0, 64, size=(n, 3)) / 63.0).T).reshape(-1)
    raise ValueError(kind)


def palette(flat, n, k, how):
    pts = flat.reshape(3, n).T
    pal = pts[rng.choice(n, size=k, replace=False)].copy()
    if how == "dups":
        pal[k // 3] = pal[k // 5]; pal[k - 1] = pal[0]                # noqa: E702
    elif how == "clump":                                 # half of the palette crowded into one corner of the space
        c = pal[0]
        pal[: k // 2] = c + rng.standard_normal((k // 2, 3)) * 1e-3 * (np.abs(c) + 0.01)
    elif how == "line":
        t = np.linspace(0, 1, k)[:, None]
        pal = pts.min(0) + t * (pts.max(0) - pts.min(0))
    return pal


bad = 0
for case in range(ncases):
    kind = str(rng.choice(["ictcp", "luv", "blobs", "dark", "quantised"]))
    how = str(rng.choice(["plain", "dups", "clump", "line"]))
    # ---- NN map, LDS-table path
    n = int(rng.integers(4_200_000, 6_000_000))
    k = int(rng.choice([16, 64, 200, 256]))
    flat = pixels(kind, n)
    pal = palette(flat, n, k, how)
    got = np.zeros(n, dtype=np.uintp)
    assert L.patolette_amd_nn_map(d(flat), n, d(np.ascontiguousarray(pal.T).reshape(-1)), k, got.ctypes.data_as(zp)) == 0
    want = ob.nn_map(flat, n, pal)
    mm = int(np.sum(got != want))
    print("case %d nn_map  %-9s %-5s n=%d k=%d: %d mismatches" % (case, kind, how, n, k, mm), flush=True)
    bad += mm != 0
    # ---- KMeans, LDS-table assignment (k a multiple of 8), a few iterations over all samples
    n = int(rng.integers(2_100_000, 3_000_000))
    k = int(rng.choice([16, 64, 128, 256]))
    flat = pixels(kind, n)
    cent = palette(flat, n, k, how)
    w = (1.0 + 3.0 * rng.random(n)) if rng.integers(0, 2) else None
    want = ob.kmeans_refine(flat, w, n, cent, 3, n)
    c = np.ascontiguousarray(cent.T).reshape(-1).copy()
    assert L.patolette_amd_kmeans_refine(d(flat), d(w) if w is not None else None, n, d(c), k, 3, n) == 0
    gotc = c.reshape(3, k).T.astype(np.float32)
    mm = int(np.sum(gotc.view(np.uint32) != want.astype(np.float32).view(np.uint32)))
    print("case %d kmeans  %-9s %-5s n=%d k=%d weighted=%d: %d floats differ" % (case, kind, how, n, k, w is not None, mm), flush=True)
    bad += mm != 0
print("FAILED" if bad else "all equal", bad)
