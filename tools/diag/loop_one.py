"""One image through both split-loop drivers: stats + timing (GPU box).  usage: loop_one.py W H K [cs] [weighted]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
from patolette_amd import _native as native
from oracle import binding as ob
L = native.lib()
w, h, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cs = int(sys.argv[4]) if len(sys.argv) > 4 else 2
weighted = len(sys.argv) > 5 and sys.argv[5] == "1"
n = w * h
flat = ob.image(n, 0)
colors = flat.reshape(3, n).T.copy()
wts = ob.weights(n, 0) if weighted else None
out = {}
for mode in (0, 1, 0, 1):
    L.patolette_amd_set_split_loop(mode)
    t = []
    for rep in range(4):
        t0 = time.perf_counter()
        ok, pal, pmap, msg = p.quantize(w, h, colors, K, dither=False, color_space=cs, tile_size=0, kmeans_niter=0, weights=wts)
        t.append(time.perf_counter() - t0)
    st = p.last_stats()
    print("mode", mode, "ok", ok, {k: st[k] for k in ("ms_gq", "ms_lq", "n_clusters", "split_evals", "split_px", "lq_rounds")}, "host-to-host ms", ["%.2f" % (1e3 * v) for v in t])
    out[mode] = (pal, pmap)
print("same palette/map:", np.array_equal(out[0][0], out[1][0]), np.array_equal(out[0][1], out[1][1]))
