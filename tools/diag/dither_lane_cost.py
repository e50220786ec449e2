"""Where a step of k_dither_lanes goes (diagnostic build: make -C patolette_amd/csrc TRACE=1; timings only, the maps of the
flagged variants are wrong): the speculative launch's time with the record load made independent of the query (flag 4), without
the exact pass (flag 8), both, and with the 32^3 grid.  usage: dither_lane_cost.py [side=4096]"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, ".")
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import numpy as np
    from patolette_amd import _native as native
    native.LIB_PATH = os.path.join(os.path.dirname(native.LIB_PATH), "trace", "libpatolette_amd.so")
    L = native.lib()
    raw = C.CDLL(native.LIB_PATH)
    side = int(sys.argv[1])
    n = side * side
    flags = int(os.environ.get("DLC_FLAGS", "0"))
    img = C.c_void_p(L.patolette_amd_malloc(3 * n * 8))
    dmap = C.c_void_p(L.patolette_amd_malloc(n))
    assert L.patolette_amd_fill_image(img, n, 7) == 0
    if os.environ.get("DK_CONTENT", "noise") == "scene":
        from tests.util import scene
        planes = np.ascontiguousarray(np.moveaxis(np.kron(scene(side // 2, side // 2, 4), np.ones((2, 2, 1))), 2, 0))
        flat = np.ascontiguousarray(planes.reshape(-1))
        assert L.patolette_amd_memcpy_h2d(img, flat.ctypes.data_as(C.c_void_p), flat.nbytes) == 0
    opts = native.QuantizationOptions(True, False, 2, 0, 512 ** 2, False)
    pal = np.zeros((256, 3), dtype=np.float64, order="F")
    code = C.c_int(0)
    raw.patolette_amd_debug_nn_flags(0)
    L.patolette_amd_device(side, side, img, None, 256, C.byref(opts), pal.ctypes.data_as(native.dp), dmap, 1, C.byref(code))
    raw.patolette_amd_debug_nn_flags(flags)
    native.profile(True)
    L.patolette_amd_device(side, side, img, None, 256, C.byref(opts), pal.ctypes.data_as(native.dp), dmap, 1, C.byref(code))
    pr = native.profile_results()
    raw.patolette_amd_debug_nn_flags(0)
    st = native.last_stats()
    print("flags %d grid %s: k_dither (speculative launch) %.3f ms; runs %d" % (flags, os.environ.get("PAMD_DITHER_GRID", "64"), pr["k_dither"]["total_ms"], st["dither_segments"]))
    sys.exit(0)
side = sys.argv[1] if len(sys.argv) > 1 else "4096"
for flags, grid in ((0, "64"), (16, "64"), (24, "64")):
    env = dict(os.environ, DLC_FLAGS=str(flags), PAMD_DITHER_GRID=grid)
    subprocess.run([sys.executable, __file__, side, "child"], env=env, timeout=300)
