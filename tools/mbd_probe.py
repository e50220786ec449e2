import sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
import patolette_amd as p
from patolette_amd import _native
L = _native.lib()
fp = C.POINTER(C.c_float)
rng = np.random.default_rng(1)
for rows in (66, 130, 258, 1026, 4096):
    cols = 4096
    img = rng.random((rows, cols), dtype=np.float32)
    out = np.zeros_like(img)
    L.patolette_amd_mbd(rows, cols, img.ctypes.data_as(fp), 3, out.ctypes.data_as(fp))
    p.profile(True)
    L.patolette_amd_mbd(rows, cols, img.ctypes.data_as(fp), 3, out.ctypes.data_as(fp))
    r = p.profile_results()["k_mbd_scan"]
    p.profile(False)
    print("rows %5d strips %3d: %.1f us per pass" % (rows, (rows - 2 + 63) // 64, 1e3 * r["total_ms"] / r["launches"]))
