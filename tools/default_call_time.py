"""The reference binding's default call (dither on, ICtCp, tile_size 512, KMeans 32 it, K = 256) through patolette_amd_u8 on one
image size, a few repetitions: stage times of the last one.  usage: default_call_time.py [width] [height] [reps] [tile]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from patolette_amd import _native as native

L = native.lib()
w = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tile = float(sys.argv[4]) if len(sys.argv) > 4 else 512.0
n = w * h
img = np.random.default_rng(77).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
opts = native.QuantizationOptions(True, False, 2, 32, 512 ** 2, False)
pal = np.zeros((256, 3), dtype=np.float64, order="F")
pal8 = np.zeros((256, 3), dtype=np.uint8)
pmap = np.zeros(n, dtype=np.uint8)
code = C.c_int(0)
for i in range(reps):
    t0 = time.perf_counter()
    L.patolette_amd_u8(w, h, img.ctypes.data_as(C.c_void_p), 3, None, C.c_double(tile), 256, C.byref(opts),
                       pal.ctypes.data_as(native.dp), pal8.ctypes.data_as(C.c_void_p), pmap.ctypes.data_as(C.c_void_p), 1, None, C.byref(code))
    dt = time.perf_counter() - t0
    assert code.value == 0, native.last_error()
    st = native.last_stats()
    print("call %d: %.3f ms  %s  runs %d repairs %d passes %d" % (i, 1e3 * dt, {k: round(v, 3) for k, v in st.items() if k.startswith("ms_") and v},
                                                                 st["dither_segments"], st["dither_repairs"], st["dither_rounds"]), flush=True)
