#!/usr/bin/env python3
"""A/B of k_nn_map_mid variants INSIDE one process, launches interleaved (a single 67 MP launch varies by +-7 %): the map stage of
the two 67 MP geometries under PAMD_NN_WAVES = 16 / 12, and -- with the diagnostic build (PAMD_LIB_DIR=trace) -- with the drains
switched off (flag 1) or nothing parked at all (flag 2): timing experiments, wrong maps.
usage: nn_ab.py [rounds=10]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from patolette_amd import _native  # noqa: E402

trace = os.environ.get("PAMD_LIB_DIR") == "trace"
if trace:
    _native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), "trace", "libpatolette_amd.so")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
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
variants = [("w16", "16", 0), ("w12", "12", 0)]
if trace:
    variants += [("w16-nodrain", "16", 1), ("w16-nopark", "16", 2)]
for name, opts, wts in (("c4km", _native.QuantizationOptions(False, False, 2, 2, n, False), None),
                        ("c4map", _native.QuantizationOptions(False, False, 1, 0, 512 ** 2, False), wt)):
    times = {v[0]: [] for v in variants}
    for r in range(rounds + 1):
        for vname, waves, flag in variants:
            os.environ["PAMD_NN_WAVES"] = waves
            if trace:
                raw.patolette_amd_debug_nn_flags(flag)
            _native.profile(True, only="k_nn_map")
            L.patolette_amd_device(w, h, img, wts, K, C.byref(opts), pal.ctypes.data_as(_native.dp), dmap, 1, C.byref(code))
            assert code.value == 0, _native.last_error()
            L.patolette_amd_synchronize()
            pr = _native.profile_results().get("k_nn_map")
            _native.profile(False)
            if r:
                times[vname].append(1e3 * pr["total_ms"] / pr["launches"])
    for vname, t in times.items():
        t = sorted(t)
        print("%-6s %-12s k_nn_map min %6.1f  median %6.1f  max %6.1f us   (median: %.3f of 8 TB/s)"
              % (name, vname, t[0], t[len(t) // 2], t[-1], 25.0 * n / (t[len(t) // 2] * 1e-6) / 8e12), flush=True)
