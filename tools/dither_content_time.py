"""Map-stage time of the dithered call on image CONTENT other than noise, for several warm-up lengths of the lane layout: smooth
analytic gradients, the same with a little noise, posterised, the tests' scene (enlarged 2 x: smoother, like a large photograph).
The speculative chains take longest to meet the true one where the errors are small.  usage: dither_content_time.py [side=4096] ["S:W,..."]"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from patolette_amd import _native
from tests.util import scene

L = _native.lib()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cuts = sys.argv[2] if len(sys.argv) > 2 else "0:-1,0:384,0:256,0:192"
n = side * side
rng = np.random.default_rng(3)
img = C.c_void_p(L.patolette_amd_malloc(3 * n * 8))
dmap = C.c_void_p(L.patolette_amd_malloc(n))
opts = _native.QuantizationOptions(True, False, 2, 0, 512 ** 2, False)
pal = np.zeros((256, 3), dtype=np.float64, order="F")
code = C.c_int(0)


def contents():
    y, x = np.mgrid[0:side, 0:side].astype(np.float32)
    sm = np.stack([0.5 + 0.5 * np.sin(x / 337.0) * np.cos(y / 253.0), (x + y) / (2.0 * side), 0.5 + 0.5 * np.cos((x - y) / 571.0)]).astype(np.float64)
    del x, y
    yield "smooth", sm
    yield "smooth+noise", np.clip(sm + 0.02 * rng.standard_normal(sm.shape), 0, 1)
    yield "posterised", np.round(sm * 7) / 7
    del sm
    sc = scene(side // 2, side // 2, 4)
    sc = np.kron(sc, np.ones((2, 2, 1)))
    yield "scene x2", np.ascontiguousarray(np.moveaxis(sc, 2, 0))
    yield "scene x2, 8-bit", np.round(np.ascontiguousarray(np.moveaxis(sc, 2, 0)) * 255) / 255


for name, planes in contents():
    flat = np.ascontiguousarray(planes.reshape(-1))
    assert L.patolette_amd_memcpy_h2d(img, flat.ctypes.data_as(C.c_void_p), flat.nbytes) == 0
    del flat, planes
    ref = None
    for cut in cuts.split(","):
        s, w = (int(v) for v in cut.split(":"))
        L.patolette_amd_dither_config(s, w)
        best = None
        for rep in range(3):
            L.patolette_amd_device(side, side, img, None, 256, C.byref(opts), pal.ctypes.data_as(C.POINTER(C.c_double)), dmap, 1, C.byref(code))
            assert code.value == 0, _native.last_error()
            st = _native.last_stats()
            best = st if best is None or st["ms_map"] < best["ms_map"] else best
        got = np.empty(n, dtype=np.uint8)
        L.patolette_amd_memcpy_d2h(got.ctypes.data_as(C.c_void_p), dmap, n)
        if ref is None:
            ref = got
        print("%-16s S=%6d warm=%4d: runs %6d repairs %6d passes %2d through %d  map stage %7.2f ms  same map as the first cut: %s" % (
            name, s, w, best["dither_segments"], best["dither_repairs"], best["dither_rounds"], best["dither_through"], best["ms_map"],
            bool(np.array_equal(got, ref))), flush=True)
L.patolette_amd_dither_config(0, -1)
