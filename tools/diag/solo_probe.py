"""Which banded images make the lane dither take solo passes (GPU box): bands of 200 rows, palette by the oracle's quantiser
from a reduced copy (fewer rows than bands)."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from patolette_amd import _native as native
from oracle import binding as ob
L = native.lib()
dp = C.POINTER(C.c_double); zp = C.POINTER(C.c_size_t)
L.patolette_amd_dither_layout(1)
for (w, h, bh, K, seed) in [(4849, 2258, 200, 10, 0), (4849, 2258, 200, 8, 1), (5413, 1913, 200, 10, 2), (4096, 2400, 200, 12, 3), (4096, 2400, 100, 8, 4), (3000, 3600, 300, 8, 5), (2048, 2048, 128, 8, 6), (2048, 2048, 64, 12, 7)]:
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w, 3))
    for y0 in range(0, h, bh):
        img[y0:y0 + bh] = rng.random(3)
    small = img[::8, ::8]
    sh, sw = small.shape[:2]
    ssrgb = np.concatenate([small[:, :, c].reshape(-1) for c in range(3)])
    ec, pal, _ = ob.patolette(sw, sh, ssrgb, None, K, dither=False, color_space=2, kmeans_niter=0)
    pal = pal[pal[:, 0] >= 0]
    if pal.shape[0] < 8:
        pal = np.vstack([pal, rng.random((8 - pal.shape[0], 3))])
    pal = ob.convert("srgb_to_rec2020", ob.planar(pal).copy()).reshape(3, -1).T.copy()
    flat = ob.convert("srgb_to_rec2020", np.concatenate([img[:, :, c].reshape(-1) for c in range(3)]))
    got = np.zeros(w * h, dtype=np.uintp)
    p = np.ascontiguousarray(pal.T).reshape(-1)
    assert L.patolette_amd_dither(flat.ctypes.data_as(dp), w, h, p.ctypes.data_as(dp), pal.shape[0], got.ctypes.data_as(zp)) == 0
    st = native.last_stats()
    print(w, h, bh, K, seed, {q: st[q] for q in ("dither_segments", "dither_repairs", "dither_rounds", "dither_through", "dither_jumps", "dither_solo")}, flush=True)
