import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
from oracle import binding as ob
from patolette_amd import _native
L = _native.lib()
dp = C.POINTER(C.c_double)
def _d(a): return a.ctypes.data_as(dp) if a is not None else None
ob.set_threads(32)
for (k, n, niter) in [(4096, 16384, 1), (4104, 16384, 1), (4096, 16384, 2), (4104, 16384, 2), (4104, 300000, 1), (4104, 300000, 2), (5000, 16384, 1), (5000,16384,2)]:
    flat = ob.convert("srgb_to_ictcp", ob.image(n, 41))
    pts = flat.reshape(3, n).T
    rng = np.random.default_rng(k + n)
    cent = pts[rng.choice(n, size=k, replace=False)].copy() + 1e-4 * rng.standard_normal((k, 3))
    want = ob.kmeans_refine(flat, None, n, cent, niter, 512 ** 2).astype(np.float32)
    c = np.ascontiguousarray(cent.T).reshape(-1).copy()
    rc = L.patolette_amd_kmeans_refine(_d(flat), None, n, _d(c), k, niter, 512 ** 2)
    got = c.reshape(3, k).T.astype(np.float32)
    bad = np.where(np.any(got.view(np.uint32) != want.view(np.uint32), axis=1))[0]
    print(k, n, niter, "rc", rc, "rows differing", len(bad), bad[:10], flush=True)
    if len(bad):
        print(got[bad[:3]], want[bad[:3]])
