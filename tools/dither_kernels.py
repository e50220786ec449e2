"""Per-kernel HIP-event times of the map stage of one dithered, device-resident call (KTIME names: k_nn_lut_build, k_dither_order,
k_dither_gather = k_dither_streams, k_dither = the speculative launch, k_dither_fix = checks + repair launches, k_dither_unpermute).
usage: [DST_CS=1] [DST_NITER=0] [DK_CONTENT=noise] [DK_WEIGHTS=1] dither_kernels.py [side=8192] [K=256]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from patolette_amd import _native

if os.environ.get("PAMD_DIAG_TRACE"):                      # the diagnostic build (make -C patolette_amd/csrc TRACE=1 STATS=1): its counters, no timings
    _native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), "trace", "libpatolette_amd.so")
if os.environ.get("PAMD_VARIANT"):                         # an experimental build (make -C patolette_amd/csrc VARIANT=name EXTRA=...)
    _native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), os.environ["PAMD_VARIANT"], "libpatolette_amd.so")
L = _native.lib()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = side * side
img = C.c_void_p(L.patolette_amd_malloc(3 * n * 8))
wt = C.c_void_p(L.patolette_amd_malloc(n * 8))
dmap = C.c_void_p(L.patolette_amd_malloc(n))
assert L.patolette_amd_fill_image(img, n, 7) == 0 and L.patolette_amd_fill_weights(wt, n, 7) == 0
content = os.environ.get("DK_CONTENT", "noise")            # noise | smooth | smooth+noise | posterised | scene (the tests' scene, enlarged 2 x)
if content != "noise":
    if content == "scene":
        from tests.util import scene
        planes = np.ascontiguousarray(np.moveaxis(np.kron(scene(side // 2, side // 2, 4), np.ones((2, 2, 1))), 2, 0))
    else:
        y, x = np.mgrid[0:side, 0:side].astype(np.float32)
        planes = np.stack([0.5 + 0.5 * np.sin(x / 337.0) * np.cos(y / 253.0), (x + y) / (2.0 * side), 0.5 + 0.5 * np.cos((x - y) / 571.0)]).astype(np.float64)
        if content == "smooth+noise":
            planes = np.clip(planes + 0.02 * np.random.default_rng(3).standard_normal(planes.shape), 0, 1)
        elif content == "posterised":
            planes = np.round(planes * 7) / 7
    flat = np.ascontiguousarray(planes.reshape(-1))
    assert L.patolette_amd_memcpy_h2d(img, flat.ctypes.data_as(C.c_void_p), flat.nbytes) == 0
cs, niter = int(os.environ.get("DST_CS", "1")), int(os.environ.get("DST_NITER", "0"))
opts = _native.QuantizationOptions(True, False, cs, niter, 512 ** 2, False)
pal = np.zeros((K, 3), dtype=np.float64, order="F")
code = C.c_int(0)
raw = C.CDLL(_native.LIB_PATH)
st8 = (C.c_ulonglong * 8)()
for rep in range(3):
    if rep == 2:
        _native.profile(True)
        if hasattr(raw, "patolette_amd_debug_dither_lane_stats"):
            raw.patolette_amd_debug_dither_lane_stats(st8, 1)
            raw.patolette_amd_debug_nn_stats(st8, 1)
    L.patolette_amd_device(side, side, img, wt if os.environ.get("DK_WEIGHTS", "1") != "0" else None, K, C.byref(opts), pal.ctypes.data_as(C.POINTER(C.c_double)), dmap, 1, C.byref(code))
    assert code.value == 0, _native.last_error()
pr = _native.profile_results()
_native.profile(False)
st = _native.last_stats()
print("%s %dx%d K=%d cs=%d: ms_map %.3f, runs %d repairs %d passes %d through %d" % (content, side, side, K, cs, st["ms_map"], st["dither_segments"], st["dither_repairs"],
                                                                               st["dither_rounds"], st["dither_through"]))
if hasattr(raw, "patolette_amd_debug_dither_lane_stats"):
    raw.patolette_amd_debug_dither_lane_stats(st8, 1)
    v = [int(x) for x in st8]
    raw.patolette_amd_debug_nn_stats(st8, 1)
    if v[0]:
        print("   the wavefront with most of them: %d steps with an exact pass, %d with a full scan" % (int(st8[6]), int(st8[7])))
        print("   lane-steps %d: all k entries (outside both grids / overflowed cell) %.4f, crowded cell (> 15) %.4f, ambiguous %.4f; wavefront-steps %d: with an exact pass %.3f, with a full scan %.3f, mean longest list %.2f"
              % (v[0], v[1] / v[0], v[2] / v[0], v[3] / v[0], v[4], v[5] / v[4], v[6] / v[4], v[7] / v[4]))
if hasattr(raw, "patolette_amd_debug_dither_lane_waves"):
    nwv = min(4096, (st["dither_segments"] + 63) // 64)
    wv = np.zeros((nwv, 4), dtype=np.uint64)
    raw.patolette_amd_debug_dither_lane_waves(wv.ctypes.data_as(C.c_void_p), nwv)
    order = np.argsort(-wv[:, 0].astype(np.int64))
    print("   wavefronts of the speculative launch by duration (100 MHz clocks): median %d; the five longest:" % int(np.median(wv[:, 0])))
    for i in order[:5]:
        print("      wavefront %4d: %7d clocks, %3d steps with an exact pass, longest lists sum %5d, %3d steps with a lane that takes all k entries" % (i, *[int(x) for x in wv[i]]))
for k_, v in pr.items():
    if "dither" in k_ or "nn_" in k_:
        print("   %-20s %8.3f ms in %d launches" % (k_, v["total_ms"], v["launches"]))
