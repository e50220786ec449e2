import sys, time, numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
import os
from patolette_amd import _native
if os.environ.get("PAMD_LIB_DIR"):                       # A/B of two builds in one gpurun call: patolette_amd/lib/<dir>/libpatolette_amd.so
    _native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), os.environ["PAMD_LIB_DIR"], "libpatolette_amd.so")
n = 2048
rng = np.random.default_rng(3)
colors = rng.random((n * n, 3))
for K in (256, 64):
    p.quantize(n, n, colors, K, dither=True, tile_size=0, kmeans_niter=0)
    t0 = time.perf_counter()
    ok = p.quantize(n, n, colors, K, dither=True, tile_size=0, kmeans_niter=0)[0]
    st = _native.last_stats()
    print("K=%d: map stage %.1f ms = %.1f ns/px" % (K, st["ms_map"], 1e6 * st["ms_map"] / (n * n)))
