"""k_convert's HIP-event time at 4096^2 (sRGB -> ICtCp and sRGB -> CIELuv) for a build variant: PAMD_VARIANT=name picks
patolette_amd/lib/<name>/libpatolette_amd.so (make -C patolette_amd/csrc VARIANT=name EXTRA=-DPAMD_CONVERT_WAVES=n)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from patolette_amd import _native

if os.environ.get("PAMD_VARIANT"):
    _native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), os.environ["PAMD_VARIANT"], "libpatolette_amd.so")
L = _native.lib()
side = 4096
n = side * side
img = C.c_void_p(L.patolette_amd_malloc(3 * n * 8))
dmap = C.c_void_p(L.patolette_amd_malloc(n))
assert L.patolette_amd_fill_image(img, n, 0) == 0
pal = np.zeros((256, 3), dtype=np.float64, order="F")
code = C.c_int(0)
for cs in (2, 1):
    opts = _native.QuantizationOptions(False, False, cs, 0, 512 ** 2, False)
    L.patolette_amd_profile_enable(1)
    for i in range(6):
        L.patolette_amd_device(side, side, img, None, 256, C.byref(opts), pal.ctypes.data_as(_native.dp), dmap, 1, C.byref(code))
        assert code.value == 0
    name = C.create_string_buffer(64)
    ms, launches, byts = C.c_double(), C.c_size_t(), C.c_double()
    for i in range(L.patolette_amd_profile_count()):
        L.patolette_amd_profile_get(i, name, C.byref(ms), C.byref(launches), C.byref(byts))
        if name.value == b"k_convert":
            print("variant %s colour space %d: k_convert %.1f us" % (os.environ.get("PAMD_VARIANT", "(product)"), cs, 1e3 * ms.value / launches.value))
    L.patolette_amd_profile_enable(0)
