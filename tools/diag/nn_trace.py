"""When the 256 blocks of k_nn_map_mid start and end (diagnostic build: make -C patolette_amd/csrc TRACE=1)."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from patolette_amd import _native
_native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), "trace", "libpatolette_amd.so")
L = _native.lib()
raw = C.CDLL(_native.LIB_PATH)
w = h = 8192; n = w * h; K = 256
img = L.patolette_amd_malloc(3 * n * 8); dmap = L.patolette_amd_malloc(n)
L.patolette_amd_fill_image(img, n, 77)
opts = _native.QuantizationOptions(False, False, 2, 0, 512 ** 2, False)
pal = np.zeros((K, 3), order="F"); code = C.c_int(0)
for rep in range(3):
    L.patolette_amd_device(w, h, img, None, K, C.byref(opts), pal.ctypes.data_as(_native.dp), dmap, 1, C.byref(code))
    L.patolette_amd_synchronize()
    t = np.zeros((256, 2), dtype=np.uint64)
    assert raw.patolette_amd_debug_nn_trace(t.ctypes.data_as(C.c_void_p)) == 0
    t = t.astype(np.float64) * 0.01
    t0 = t[:, 0].min()
    st, en = t[:, 0] - t0, t[:, 1] - t0
    q = np.percentile(en, [0, 10, 50, 90, 100])
    print("starts: max %.1f us | ends: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f us | mean busy %.1f" % (st.max(), *q, (en - st).mean()))
    # by XCD (blocks are dealt round-robin over the 8 XCDs)
    print("  end by block%8:", " ".join("%.0f" % en[i::8].mean() for i in range(8)))
