"""k_hist with and without its LDS atomics (diagnostic build; PAMD_HIST_NOATOMIC=1 gives wrong results: timing only)."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from patolette_amd import _native
_native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), "trace", "libpatolette_amd.so")
L = _native.lib()
w = h = 4096; n = w * h; K = 256
img = L.patolette_amd_malloc(3 * n * 8); dmap = L.patolette_amd_malloc(n)
L.patolette_amd_fill_image(img, n, 0)
opts = _native.QuantizationOptions(False, False, 2, 0, 512 ** 2, False)
pal = np.zeros((K, 3), order="F"); code = C.c_int(0)
L.patolette_amd_device(w, h, img, None, K, C.byref(opts), pal.ctypes.data_as(_native.dp), dmap, 1, C.byref(code))
_native.profile(True)
for _ in range(3):
    L.patolette_amd_device(w, h, img, None, K, C.byref(opts), pal.ctypes.data_as(_native.dp), dmap, 1, C.byref(code))
L.patolette_amd_synchronize()
pr = _native.profile_results()
for k in ("k_hist_lq", "k_minmax", "k_scatter_cov"):
    r = pr[k]; print(k, "avg %.1f us over %d launches, %.0f GB/s" % (1e3 * r["total_ms"] / r["launches"], r["launches"], r["bytes"] / r["total_ms"] / 1e6))
print(_native.last_stats()["split_evals"], code.value)
