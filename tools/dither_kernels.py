"""Per-kernel HIP-event times of the map stage of one dithered, device-resident call (KTIME names: k_nn_lut_build, k_dither_order,
k_dither_gather = k_dither_streams, k_dither = the speculative launch, k_dither_fix = checks + repair launches, k_dither_unpermute).
usage: [DST_CS=1] [DST_NITER=0] dither_kernels.py [side=8192] [K=256]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from patolette_amd import _native

L = _native.lib()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = side * side
img = C.c_void_p(L.patolette_amd_malloc(3 * n * 8))
wt = C.c_void_p(L.patolette_amd_malloc(n * 8))
dmap = C.c_void_p(L.patolette_amd_malloc(n))
assert L.patolette_amd_fill_image(img, n, 7) == 0 and L.patolette_amd_fill_weights(wt, n, 7) == 0
cs, niter = int(os.environ.get("DST_CS", "1")), int(os.environ.get("DST_NITER", "0"))
opts = _native.QuantizationOptions(True, False, cs, niter, 512 ** 2, False)
pal = np.zeros((K, 3), dtype=np.float64, order="F")
code = C.c_int(0)
for rep in range(3):
    if rep == 2:
        _native.profile(True)
    L.patolette_amd_device(side, side, img, wt, K, C.byref(opts), pal.ctypes.data_as(C.POINTER(C.c_double)), dmap, 1, C.byref(code))
    assert code.value == 0, _native.last_error()
pr = _native.profile_results()
_native.profile(False)
st = _native.last_stats()
print("%dx%d K=%d cs=%d: ms_map %.3f, runs %d repairs %d passes %d through %d" % (side, side, K, cs, st["ms_map"], st["dither_segments"], st["dither_repairs"],
                                                                               st["dither_rounds"], st["dither_through"]))
for k_, v in pr.items():
    if "dither" in k_ or "nn_" in k_:
        print("   %-20s %8.3f ms in %d launches" % (k_, v["total_ms"], v["launches"]))
