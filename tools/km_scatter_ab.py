#!/usr/bin/env python3
"""k_km_scatter (pairs of records per store) against k_km_scatter_lds (1024-sample batches sorted in LDS) on the 67 M-sample KMeans of
c4km, one variant per process (PAMD_KM_LDS_SORT=0/1): launch times of the KMeans kernels and a checksum of palette and map (the
sorted copy must hold the same records in the same places: the centroid chains are sequential).  usage: km_scatter_ab.py [steps]"""
import ctypes as C
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from patolette_amd import _native  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L = _native.lib()
w = h = 8192
n, K = w * h, 256
img = L.patolette_amd_malloc(3 * n * 8)
wt = L.patolette_amd_malloc(n * 8)
dmap = L.patolette_amd_malloc(n)
assert L.patolette_amd_fill_image(img, n, 77) == 0 and L.patolette_amd_fill_weights(wt, n, 77) == 0
pal = np.zeros((K, 3), dtype=np.float64, order="F")
code = C.c_int(0)
for name, wts in (("unweighted", None), ("weighted", wt)):
    opts = _native.QuantizationOptions(False, False, 2, 4, n, False)
    for i in range(steps + 1):
        if i == 1:
            _native.profile(True)
        L.patolette_amd_device(w, h, img, wts, K, C.byref(opts), pal.ctypes.data_as(_native.dp), dmap, 1, C.byref(code))
        assert code.value == 0, _native.last_error()
    L.patolette_amd_synchronize()
    pr = _native.profile_results()
    _native.profile(False)
    m8 = np.empty(n, dtype=np.uint8)
    L.patolette_amd_memcpy_d2h(m8.ctypes.data_as(C.c_void_p), dmap, n)
    print("%-10s LDS_SORT=%s  " % (name, os.environ.get("PAMD_KM_LDS_SORT", "default")) +
          "  ".join("%s %.1f us" % (q, 1e3 * pr[q]["total_ms"] / pr[q]["launches"]) for q in ("k_km_assign", "k_km_scatter", "k_km_update") if q in pr) +
          "   palette crc %08x map crc %08x" % (zlib.crc32(pal.tobytes()), zlib.crc32(m8.tobytes())), flush=True)
