import sys, numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
from patolette_amd import _native
n = 4096
rng = np.random.default_rng(5)
for frac in [float(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,0.1,0.3,0.6".split(","))]:
    colors = rng.random((n * n, 3))
    k = int(frac * n * n)
    colors[:k] = np.array([0.1, 0.2, 0.7]) + 0.004 * rng.standard_normal((k, 3))     # a dominant, slightly noisy colour
    p.quantize(n, n, colors, 256, dither=False, tile_size=0)
    ok = p.quantize(n, n, colors, 256, dither=False, tile_size=0)[0]
    st = _native.last_stats()
    print("dominant %.0f%%: total %.2f ms, kmeans %.2f ms, lq %.2f" % (100 * frac, st["ms_total"] - st["ms_upload"] - st["ms_download"], st["ms_kmeans"], st["ms_lq"]))
