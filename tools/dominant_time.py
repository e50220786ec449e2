"""KMeans refinement when ONE colour dominates the image (tools/README.md): a slightly noisy colour covering a fraction of a
4096 x 4096 noise image; total, per-stage and per-kernel device times.   python tools/dominant_time.py [fractions]"""
import sys
import numpy as np
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
    p.profile(True)
    ok = p.quantize(n, n, colors, 256, dither=False, tile_size=0)[0]
    prof = p.profile_results()
    p.profile(False)
    st = _native.last_stats()
    km = {k2: (round(v["total_ms"], 2), v["launches"]) for k2, v in prof.items() if k2.startswith("k_km")}
    print("dominant %.0f%%: total %.2f ms, kmeans %.2f ms, lq %.2f | %s" % (100 * frac, st["ms_total"] - st["ms_upload"] - st["ms_download"], st["ms_kmeans"], st["ms_lq"], km))
