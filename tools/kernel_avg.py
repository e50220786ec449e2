"""Average launch time of the kernels of one bench configuration over several fully timed steps (tools/README.md): the bench
line's per-kernel table comes from ONE step, and a single launch of the 67 MP kernels varies by +-6 %.
python tools/kernel_avg.py [config] [steps] [kernel-name-prefixes ...]"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
import bench
from patolette_amd import _native as native
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c4km"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
want = sys.argv[3:] or ["k_nn_map", "k_km_assign"]
L = native.lib()
width, height, K, cs, niter, max_samples, dither, weighted, _ = bench.CONFIGS[cfgname]
n = width * height
img = L.patolette_amd_malloc(3 * n * 8)
dmap = L.patolette_amd_malloc(n)
assert L.patolette_amd_fill_image(img, n, 77) == 0
opts = native.QuantizationOptions(dither, False, cs, niter, max_samples, False)
pal = np.zeros((K, 3), dtype=np.float64, order="F")
code = C.c_int(0)
def one():
    L.patolette_amd_device(width, height, img, None, K, C.byref(opts), pal.ctypes.data_as(native.dp), dmap, 1, C.byref(code))
    assert code.value == 0
one()
native.profile(True)
for _ in range(steps):
    one()
L.patolette_amd_synchronize()
prof = native.profile_results()
native.profile(False)
print(" ".join("%s %.2f us x%d" % (k, 1e3 * v["total_ms"] / v["launches"], v["launches"]) for k, v in sorted(prof.items()) if any(k.startswith(w) for w in want)))
