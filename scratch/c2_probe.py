import sys, time, ctypes as C, numpy as np
sys.path.insert(0, ".")
from patolette_amd import _native
L = _native.lib()
w, h, K = 1920, 1080, 256
n = w * h
img = L.patolette_amd_malloc(3 * n * 8); L.patolette_amd_fill_image(img, n, 1)
dmap = L.patolette_amd_malloc(n)
opts = _native.QuantizationOptions(False, False, 2, 0, 512 ** 2, False)
pal = np.zeros((K, 3), order="F"); code = C.c_int(0)
for it in range(8):
    t = time.perf_counter()
    L.patolette_amd_device(w, h, img, None, K, C.byref(opts), pal.ctypes.data_as(_native.dp), dmap, 1, C.byref(code))
    dt = (time.perf_counter() - t) * 1e3
    st = _native.last_stats()
    print("call %d wall %.2f ms  stats total %.2f (cv %.2f gq %.2f lq %.2f map %.2f) rounds %d evals %d" % (it, dt, st["ms_total"], st["ms_convert"], st["ms_gq"], st["ms_lq"], st["ms_map"], st["lq_rounds"], st["split_evals"]))
