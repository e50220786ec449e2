import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
from oracle import binding as ob
import patolette_amd as p
ob.set_threads(32)
for (w, h, K, niter) in [(128, 128, 5000, 0), (128, 128, 5000, 1), (128, 128, 5000, 2), (128, 128, 4000, 2), (640, 480, 5000, 3)]:
    n = w * h
    flat = ob.image(n, 90)
    colors = ob.unplanar(flat, n)
    ok, pal, pmap, msg = p.quantize(w, h, colors, K, dither=False, color_space=2, tile_size=0, kmeans_niter=niter, kmeans_max_samples=512 ** 2)
    ec, pal_o, pmap_o = ob.patolette(w, h, flat, None, K, dither=False, color_space=2, kmeans_niter=niter, kmeans_max_samples=512 ** 2)
    bad = np.where(np.any(np.abs(pal - pal_o) > 1e-9, axis=1))[0]
    print(w, h, K, niter, ok, ec, "rows differing", len(bad), bad[:10], "map mism", int(np.sum(pmap != pmap_o)), p.last_stats().get("n_clusters"), flush=True)
    if len(bad):
        print(pal[bad[:3]], pal_o[bad[:3]])
