import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from oracle import binding as ob
import patolette_amd as p
w, h, K = 512, 384, 16
n = w * h
for cs in (2, 1):
    for amp in (1e-2, 1e-3, 3e-4, 1e-4, 3e-5, 1e-5, 3e-6):
        flat = np.clip(np.repeat([0.6, 0.35, 0.2], n) + amp * (ob.image(n, 91) - 0.5), 0.0, 1.0)
        colors = ob.unplanar(flat, n)
        ok, pal, pmap, msg = p.quantize(w, h, colors, K, dither=False, color_space=cs, tile_size=0, kmeans_niter=0)
        ec, pal_o, pmap_o = ob.patolette(w, h, flat, None, K, dither=False, color_space=cs, kmeans_niter=0)
        print("guard", os.environ.get("PAMD_ROOT_MOMENTS_GUARD", "1"), "cs", cs, "amp", amp, "pal maxdiff %.3g" % np.max(np.abs(pal - pal_o)), "map mism", int(np.sum(pmap != pmap_o)), flush=True)
