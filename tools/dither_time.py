import sys, time, numpy as np, ctypes as C
sys.path.insert(0, ".")
from patolette_amd import _native
from oracle import binding as ob
L = _native.lib()
w, h, K = 2048, 2048, 256
n = w * h
flat = ob.convert("srgb_to_rec2020", ob.image(n, 3))
pal = ob.convert("srgb_to_rec2020", ob.image(K, 77)).reshape(3, K).T.copy()
p = np.ascontiguousarray(pal.T).reshape(-1)
out = np.zeros(n, dtype=np.uintp)
dp = C.POINTER(C.c_double); zp = C.POINTER(C.c_size_t)
for it in range(2):
    t = time.time(); L.patolette_amd_dither(flat.ctypes.data_as(dp), w, h, p.ctypes.data_as(dp), K, out.ctypes.data_as(zp)); dt = time.time() - t
    print("dither %dx%d K=%d: %.1f ms host-to-host -> %.1f ns/px" % (w, h, K, dt * 1e3, dt * 1e9 / n))
want = ob.dither_prefix(flat, w, h, pal, 300000)
vis = want != 0xFFFF
print("prefix equal:", np.array_equal(out[vis], want[vis]), int(vis.sum()))
