"""Phase timestamps inside k_km_assign_sort / k_km_update_lists (diagnostic build: make -C patolette_amd/csrc TRACE=1)."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from patolette_amd import _native
_native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), "trace", "libpatolette_amd.so")
L = _native.lib()
raw = C.CDLL(_native.LIB_PATH)
w = h = 4096; n = w * h; K = 256
img = L.patolette_amd_malloc(3 * n * 8); dmap = L.patolette_amd_malloc(n)
L.patolette_amd_fill_image(img, n, 0)
opts = _native.QuantizationOptions(False, False, 2, 32, 512 ** 2, False)
pal = np.zeros((K, 3), order="F"); code = C.c_int(0)
for _ in range(3):
    L.patolette_amd_device(w, h, img, None, K, C.byref(opts), pal.ctypes.data_as(_native.dp), dmap, 1, C.byref(code))
L.patolette_amd_synchronize()
print("stats", {k: round(v, 3) for k, v in _native.last_stats().items() if k.startswith("ms_")})
t = np.zeros((2, 256, 32), dtype=np.uint64)
assert raw.patolette_amd_debug_km_trace(t.ctypes.data_as(C.c_void_p)) == 0
cyc = (t[0, :, 25].astype(np.int64) - t[0, :, 24].astype(np.int64)).astype(np.float64)
wall = (t[0, :, 8].astype(np.int64) - t[0, :, 2].astype(np.int64)).astype(np.float64) * 0.01
print("scan of wave 0: cycles median %.0f, wall %.2f us -> %.0f MHz" % (np.median(cyc), np.median(wall), np.median(cyc / np.maximum(wall, 1e-9))))
t = t.astype(np.float64) * 0.01            # 100 MHz -> us
for kern, name, nph in ((0, "assign_sort", 6), (1, "update_lists", 4)):
    a = t[kern]
    t0 = a[:, 0].min()
    print(name, "block start spread (us): min 0 median %.2f max %.2f" % (np.median(a[:, 0]) - t0, a[:, 0].max() - t0))
    for ph in range(1, nph):
        d = a[:, ph] - a[:, 0]
        print("  phase %d since block start: median %.2f max %.2f | since kernel start: max %.2f" % (ph, np.median(d), d.max(), (a[:, ph] - t0).max()))
    nw = 16 if kern == 0 else 4
    wv = a[:, 8:8 + nw] - a[:, [0]]
    print("  per-wave mark since block start: median %.2f min %.2f max %.2f" % (np.median(wv), wv.min(), wv.max()))
