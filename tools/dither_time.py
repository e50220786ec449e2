import sys, time, numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
from patolette_amd import _native
n = 2048
rng = np.random.default_rng(3)
colors = rng.random((n * n, 3))
for K in (256, 64):
    p.quantize(n, n, colors, K, dither=True, tile_size=0, kmeans_niter=0)
    t0 = time.perf_counter()
    ok = p.quantize(n, n, colors, K, dither=True, tile_size=0, kmeans_niter=0)[0]
    st = _native.last_stats()
    print("K=%d: map stage %.1f ms = %.1f ns/px" % (K, st["ms_map"], 1e6 * st["ms_map"] / (n * n)))
