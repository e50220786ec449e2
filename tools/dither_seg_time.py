"""Segment-parallel dither: time of the map stage of a dithered patolette_amd_device call for several cuts of the curve
(runs x warm-up), device-resident image.  usage: [DST_CS=1] [DST_NITER=0] dither_seg_time.py [side=8192] [K=256] ["S:W,S:W,..."]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from patolette_amd import _native

L = _native.lib()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cuts = sys.argv[3] if len(sys.argv) > 3 else "0:-1,1024:1024,2048:1024,2048:512,2560:1024,4096:1024,4096:512,2048:2048,2048:256"
n = side * side
img = C.c_void_p(L.patolette_amd_malloc(3 * n * 8))
wt = C.c_void_p(L.patolette_amd_malloc(n * 8))
dmap = C.c_void_p(L.patolette_amd_malloc(n))
assert L.patolette_amd_fill_image(img, n, 7) == 0 and L.patolette_amd_fill_weights(wt, n, 7) == 0
import os
cs, niter = int(os.environ.get("DST_CS", "1")), int(os.environ.get("DST_NITER", "0"))      # colour space (1 CIELuv, 2 ICtCp, 0 sRGB), KMeans iterations
opts = _native.QuantizationOptions(True, False, cs, niter, 512 ** 2, False)
pal = np.zeros((K, 3), dtype=np.float64, order="F")
code = C.c_int(0)
ref = None
for cut in cuts.split(","):
    s, w = (int(v) for v in cut.split(":"))
    L.patolette_amd_dither_config(s, w)
    best = None
    for rep in range(3):
        L.patolette_amd_device(side, side, img, wt, K, C.byref(opts), pal.ctypes.data_as(C.POINTER(C.c_double)), dmap, 1, C.byref(code))
        assert code.value == 0, _native.last_error()
        st = _native.last_stats()
        best = st if best is None or st["ms_map"] < best["ms_map"] else best
    got = np.empty(n, dtype=np.uint8)
    L.patolette_amd_memcpy_d2h(got.ctypes.data_as(C.c_void_p), dmap, n)
    if ref is None:
        ref = got
    print("S=%5d warm=%5d: runs %5d repairs %4d rounds %d  map stage %8.2f ms = %.3f ns/px   same map as first cut: %s"
          % (s, w, best["dither_segments"], best["dither_repairs"], best["dither_rounds"], best["ms_map"], 1e6 * best["ms_map"] / n,
             bool(np.array_equal(got, ref))), flush=True)
