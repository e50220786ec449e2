"""Wall time of patolette() with dithering through a given build of the library (raw ctypes: old builds lack newer symbols).
usage: dither_ab.py <lib dir under patolette_amd/lib or ''>"""
import sys, os, time, ctypes as C, numpy as np
sub = sys.argv[1] if len(sys.argv) > 1 else ""
path = os.path.join("patolette_amd", "lib", sub, "libpatolette_amd.so")
L = C.CDLL(os.path.abspath(path))
class Q(C.Structure):
    _fields_ = [("dither", C.c_bool), ("palette_only", C.c_bool), ("color_space", C.c_int), ("kmeans_niter", C.c_int), ("kmeans_max_samples", C.c_size_t), ("verbose", C.c_bool)]
n = 2048
rng = np.random.default_rng(3)
data = np.asfortranarray(rng.random((n * n, 3)))
for K in (256, 64):
    pal = np.zeros((K, 3), order="F"); pm = np.zeros(n * n, dtype=np.uintp); code = C.c_int(0)
    o = Q(True, False, 2, 0, 512 ** 2, False)
    for rep in range(2):
        t0 = time.perf_counter()
        L.patolette(C.c_size_t(n), C.c_size_t(n), data.ctypes.data_as(C.c_void_p), None, C.c_size_t(K), C.byref(o), pal.ctypes.data_as(C.c_void_p), pm.ctypes.data_as(C.c_void_p), C.byref(code))
        dt = time.perf_counter() - t0
    print("%s K=%d: %.1f ms total = %.1f ns/px (exit %d)" % (sub or "current", K, dt * 1e3, 1e9 * dt / (n * n), code.value))
