#!/usr/bin/env python3
"""What the drains of k_nn_map_mid meet on the two 67 MP geometries (diagnostic build: make -C patolette_amd/csrc TRACE=1):
drains, pixels drained, drains that meet a list of more than four / eight entries, pixels parked as ambiguous by the f32 pass."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from patolette_amd import _native  # noqa: E402

_native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), "trace", "libpatolette_amd.so")
L = _native.lib()
raw = C.CDLL(_native.LIB_PATH)
w = h = 8192
n, K = w * h, 256
img = L.patolette_amd_malloc(3 * n * 8)
wt = L.patolette_amd_malloc(n * 8)
dmap = L.patolette_amd_malloc(n)
assert L.patolette_amd_fill_image(img, n, 77) == 0 and L.patolette_amd_fill_weights(wt, n, 77) == 0
pal = np.zeros((K, 3), dtype=np.float64, order="F")
code = C.c_int(0)
st = (C.c_ulonglong * 8)()
for name, opts, wts in (("c4km", _native.QuantizationOptions(False, False, 2, 2, n, False), None),
                        ("c4map", _native.QuantizationOptions(False, False, 1, 0, 512 ** 2, False), wt)):
    raw.patolette_amd_debug_nn_stats(st, 1)
    L.patolette_amd_device(w, h, img, wts, K, C.byref(opts), pal.ctypes.data_as(_native.dp), dmap, 1, C.byref(code))
    assert code.value == 0, _native.last_error()
    L.patolette_amd_synchronize()
    raw.patolette_amd_debug_nn_stats(st, 1)
    v = list(st)
    print("%-6s drains %d, pixels drained %d (%.2f %% of the image, %.1f per drain), drains with a list > 4: %d, > 8: %d (%.1f %%; %d pixels), ambiguous in f32: %d (%.3f %%)"
          % (name, v[0], v[1], 100.0 * v[1] / n, v[1] / max(1, v[0]), v[2], v[3], 100.0 * v[3] / max(1, v[0]), v[4], v[5], 100.0 * v[5] / n), flush=True)
