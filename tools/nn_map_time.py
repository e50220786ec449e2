#!/usr/bin/env python3
"""Mean launch time of the palette-map kernel on the two 67 MP geometries it is judged on -- `c4km` (ICtCp + KMeans palette) and
`c4map` (BASELINE configs[3]: CIELuv + weights, no KMeans) -- and a checksum of the index map, for the library variant the
environment selects (PAMD_NN_WAVES, PAMD_LIB_DIR): python tools/nn_map_time.py [steps]"""
import ctypes as C
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from patolette_amd import _native  # noqa: E402

if os.environ.get("PAMD_LIB_DIR"):
    _native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), os.environ["PAMD_LIB_DIR"], "libpatolette_amd.so")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = _native.lib()
w = h = 8192
n, K = w * h, 256
img = L.patolette_amd_malloc(3 * n * 8)
wt = L.patolette_amd_malloc(n * 8)
dmap = L.patolette_amd_malloc(n)
assert L.patolette_amd_fill_image(img, n, 77) == 0 and L.patolette_amd_fill_weights(wt, n, 77) == 0
pal = np.zeros((K, 3), dtype=np.float64, order="F")
code = C.c_int(0)
for name, opts, wts in (("c4km", _native.QuantizationOptions(False, False, 2, 2, n, False), None),
                        ("c4map", _native.QuantizationOptions(False, False, 1, 0, 512 ** 2, False), wt)):
    for i in range(steps + 1):
        if i == 1:
            _native.profile(True, only="k_nn_map")
        L.patolette_amd_device(w, h, img, wts, K, C.byref(opts), pal.ctypes.data_as(_native.dp), dmap, 1, C.byref(code))
        assert code.value == 0, _native.last_error()
    L.patolette_amd_synchronize()
    r = _native.profile_results().get("k_nn_map")
    _native.profile(False)
    m8 = np.empty(n, dtype=np.uint8)
    L.patolette_amd_memcpy_d2h(m8.ctypes.data_as(C.c_void_p), dmap, n)
    us = 1e3 * r["total_ms"] / r["launches"]
    print("%-6s k_nn_map %7.1f us x%d  %.3f of 8 TB/s   map crc %08x  (PAMD_NN_WAVES=%s)"
          % (name, us, r["launches"], 25.0 * n / (us * 1e-6) / 8e12, zlib.crc32(m8.tobytes()), os.environ.get("PAMD_NN_WAVES", "default")), flush=True)
