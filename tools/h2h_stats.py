import sys, time, numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
from patolette_amd import _native
rng = np.random.default_rng(0)
img8 = rng.integers(0, 256, size=(4096, 4096, 3), dtype=np.uint8)
fcol = np.asfortranarray(img8.reshape(-1, 3).astype(np.float64) / 255)
r = None
for it in range(4):
    r = None
    t = time.time(); r = p.quantize(4096, 4096, fcol, 256, dither=False, tile_size=0); a = time.time() - t
    st = _native.last_stats()
    print("wall %.1f ms  upload %.2f  device %.2f  download %.2f  total %.2f" % (a * 1e3, st["ms_upload"], st["ms_total"] - st["ms_upload"] - st["ms_download"], st["ms_download"], st["ms_total"]))
