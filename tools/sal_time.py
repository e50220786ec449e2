import numpy as np, time, sys
sys.path.insert(0, '.')
import patolette_amd as p
from patolette_amd import _native
from tests.util import scene
rows, cols = 4096, 4096
img = scene(rows, cols, 1)
colors = img.reshape(-1, 3)
p.profile(True)
for it in range(2):
    t = time.time(); w = p.saliency_weights(cols, rows, colors, 512); dt = time.time() - t
    print("saliency_weights host-to-host %.1f ms" % (dt * 1e3), w.min(), w.max())
for k, v in sorted(p.profile_results().items(), key=lambda kv: -kv[1]["total_ms"]):
    print("%-18s %8.3f ms  %4d launches  %.1f GB/s" % (k, v["total_ms"], v["launches"], v["bytes"] / v["total_ms"] / 1e6))
p.profile(False)
# exactness of the scans at full size against the C oracle
from oracle import binding as ob
import ctypes as C
img32 = img.mean(axis=2).astype(np.float32)
out = np.zeros_like(img32)
fp = C.POINTER(C.c_float)
t = time.time(); rc = _native.lib().patolette_amd_mbd(rows, cols, img32.ctypes.data_as(fp), 3, out.ctypes.data_as(fp)); print("gpu mbd h2h ms", (time.time() - t) * 1e3)
t = time.time(); want = ob.mbd(img32, 3); print("cpu mbd ms", (time.time() - t) * 1e3)
print("mbd 4096^2 bit-exact:", np.array_equal(out.view(np.uint32), want.view(np.uint32)))
