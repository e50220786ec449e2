"""Why is the dither stage slower behind saliency weights?  Same 4096^2 8-bit noise image, two palettes (the default call's with
tile_size 0 and with tile_size 512), the dither stage alone on each (patolette_amd_dither, host to host: only the stats matter),
plus the palettes' extents in weighted Rec2020.  Diagnostic; uses the oracle's conversions for the inputs."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import binding as ob
from patolette_amd import _native as native

import os
if os.environ.get("PAMD_DIAG_TRACE"):                      # the diagnostic build (make -C patolette_amd/csrc TRACE=1 STATS=1): counters, no timings
    native.LIB_PATH = os.path.join(os.path.dirname(native.LIB_PATH), "trace", "libpatolette_amd.so")
L = native.lib()
w = h = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = w * h
img = np.random.default_rng(77).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
dp, zp = C.POINTER(C.c_double), C.POINTER(C.c_size_t)
pals = {}
for tile in (0.0, 512.0):
    opts = native.QuantizationOptions(True, False, 2, 32, 512 ** 2, False)
    pal = np.zeros((256, 3), dtype=np.float64, order="F")
    pal8 = np.zeros((256, 3), dtype=np.uint8)
    pmap = np.zeros(n, dtype=np.uint8)
    code = C.c_int(0)
    for _ in range(2):
        L.patolette_amd_u8(w, h, img.ctypes.data_as(C.c_void_p), 3, None, C.c_double(tile), 256, C.byref(opts),
                           pal.ctypes.data_as(dp), pal8.ctypes.data_as(C.c_void_p), pmap.ctypes.data_as(C.c_void_p), 1, None, C.byref(code))
    assert code.value == 0
    st = native.last_stats()
    print("tile %4d: ms_map %.3f  hist of map usage: min %d max %d" % (tile, st["ms_map"], np.bincount(pmap, minlength=256).min(), np.bincount(pmap, minlength=256).max()))
    pals[tile] = np.array(pal)
flat = np.concatenate([(img[:, :, c].reshape(-1) / 255.0) for c in range(3)])
rec = ob.convert("srgb_to_rec2020", flat)
for tile, pal in pals.items():
    pr = ob.convert("srgb_to_rec2020", ob.planar(pal).copy()).reshape(3, -1).T.copy()
    print("tile %4d palette (Rec2020): min %s max %s" % (tile, pr.min(0).round(4), pr.max(0).round(4)))
    got = np.zeros(n, dtype=np.uintp)
    p = np.ascontiguousarray(pr.T).reshape(-1)
    assert L.patolette_amd_dither(rec.ctypes.data_as(dp), w, h, p.ctypes.data_as(dp), 256, got.ctypes.data_as(zp)) == 0
    native.profile(True)
    assert L.patolette_amd_dither(rec.ctypes.data_as(dp), w, h, p.ctypes.data_as(dp), 256, got.ctypes.data_as(zp)) == 0
    pr_ = native.profile_results()
    native.profile(False)
    st = native.last_stats()
    print("   dither alone: runs %d repairs %d passes %d; kernels (ms): %s" % (st["dither_segments"], st["dither_repairs"], st["dither_rounds"],
          {k_: round(v["total_ms"], 3) for k_, v in pr_.items()}))
    try:
        raw = C.CDLL(native.LIB_PATH)
        st8 = (C.c_ulonglong * 8)()
        raw.patolette_amd_debug_dither_lane_stats(st8, 1)
        assert L.patolette_amd_dither(rec.ctypes.data_as(dp), w, h, p.ctypes.data_as(dp), 256, got.ctypes.data_as(zp)) == 0
        raw.patolette_amd_debug_dither_lane_stats(st8, 1)
        v = [int(x) for x in st8]
        if v[0]:
            print("   lane-steps %d: outside the grid %.4f, long-list cell %.4f, ambiguous %.4f; wavefront-steps %d: with an exact pass %.3f, with a full scan %.3f, mean trips %.2f"
                  % (v[0], v[1] / v[0], v[2] / v[0], v[3] / v[0], v[4], v[5] / v[4], v[6] / v[4], v[7] / v[4]))
    except AttributeError:
        pass                                               # (not a STATS build)
    # the quantisation error the chain carries: how far do the queries stray from the palette's box?
    err = rec.reshape(3, -1).T[::97] - pr[got[::97]]
    print("   |pixel - chosen entry| (Rec2020, every 97th pixel): mean %s  max %s" % (np.abs(err).mean(0).round(4), np.abs(err).max(0).round(3)))
