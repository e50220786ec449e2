"""Per-kernel KMeans times on the posterised scene of bench.py's content extras (tools/README.md), default update and the
order-free option.   python tools/content_kmeans.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
from patolette_amd import _native
from tests.util import scene
n = 4096
img = scene(n, n, 4)
post = (np.round(img * 7.0) / 7.0).reshape(-1, 3)
for mode in (0, 1):
    p.set_kmeans_update(mode)
    p.quantize(n, n, post, 256, dither=False, tile_size=0)
    p.profile(True)
    p.quantize(n, n, post, 256, dither=False, tile_size=0)
    prof = p.profile_results()
    p.profile(False)
    st = _native.last_stats()
    km = {k2: (round(v["total_ms"], 2), v["launches"]) for k2, v in prof.items() if k2.startswith("k_km")}
    print("posterised, update %d: kmeans %.2f ms | %s" % (mode, st["ms_kmeans"], km))
p.set_kmeans_update(0)
