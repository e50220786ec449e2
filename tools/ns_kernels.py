#!/usr/bin/env python3
"""Mean launch time of the two north-star kernels over several c4km steps (tools/README.md): python tools/ns_kernels.py [steps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from patolette_amd import _native  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L = _native.lib()
w = h = 8192
n, K = w * h, 256
img = L.patolette_amd_malloc(3 * n * 8)
dmap = L.patolette_amd_malloc(n)
assert L.patolette_amd_fill_image(img, n, 77) == 0
opts = _native.QuantizationOptions(False, False, 2, 8, n, False)
pal = np.zeros((K, 3), dtype=np.float64, order="F")
code = C.c_int(0)
for i in range(steps + 1):
    if i == 1:
        _native.profile(True)
    L.patolette_amd_device(w, h, img, None, K, C.byref(opts), pal.ctypes.data_as(_native.dp), dmap, 1, C.byref(code))
    assert code.value == 0
L.patolette_amd_synchronize()
pr = _native.profile_results()
for name, by in (("k_km_assign", 16.0), ("k_nn_map", 25.0), ("k_km_scatter", 29.0), ("k_km_update", 16.0), ("k_km_lut_build", 0), ("k_nn_lut_build", 0)):
    r = pr.get(name)
    if r:
        us = 1e3 * r["total_ms"] / r["launches"]
        print("%-16s %8.1f us  x%-3d %s" % (name, us, r["launches"], ("%.3f of 8 TB/s" % (by * n / (us * 1e-6) / 8e12)) if by else ""))
