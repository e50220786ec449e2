#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the build container only).

Sources of truth, per file:
  color_ref.npz    outputs of the REFERENCE's own colour code: oracle/_ref/libref_color.so =
                   /root/reference/lib/src/color/*.c + lib/src/array/*.c compiled as they lie.
  eigen_lapack.npz outputs of LAPACK dsyev('V','L') from the OpenBLAS 0.3.28 that ships in this
                   image (scipy.linalg.lapack.dsyev) -- the third-party routine the reference
                   calls at lib/src/math/eigen.c:105-136.
  kmeans_ref.npz   outputs of the REFERENCE's vendored+patched faiss: oracle/_ref/libref_faiss.so
                   (faiss_kmeans_clustering through lib/faiss/c_api, AVX2 flavour).
  pipeline_oracle.npz  outputs of OUR ORACLE (oracle/patolette_oracle.c) for GQ/LQ/dither/end-to-end
                   cases.  The reference's quantize/, palette/nearest.c, dither/ and patolette.c need
                   <cblas.h> and <flann/flann.h>, absent here, so no stand-in-free reference build
                   exists for them: these are REGRESSION vectors (parity unpinned by a reference
                   build).  If the survey session's throw-away shimmed build is still present at
                   /tmp/probe it is also run and agreement is recorded in `probe_equal` -- informative
                   only, that build used header/FLANN stand-ins.

Inputs are regenerated from the in-repo splitmix64 generator (orc_fill_*), so only parameters
and outputs are stored.  Nothing from /root/reference is copied: fixtures are data.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF_COLOR = os.path.join(ROOT, "oracle", "_ref", "libref_color.so")
REF_FAISS = os.path.join(ROOT, "oracle", "_ref", "libref_faiss.so")


# ----------------------------------------------------------------------------- colour
class M2(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_double)), ("rows", C.c_size_t), ("cols", C.c_size_t)]


REF_CONV = {
    "srgb_to_ictcp": "patolette__COLOR_sRGB_Matrix_to_ICtCp_Matrix",
    "srgb_to_cieluv": "patolette__COLOR_sRGB_Matrix_to_CIELuv_Matrix",
    "srgb_to_rec2020": "patolette__COLOR_sRGB_Matrix_to_Linear_Rec2020_Matrix",
    "ictcp_to_rec2020": "patolette__COLOR_ICtCp_Matrix_to_Linear_Rec2020_Matrix",
    "cieluv_to_rec2020": "patolette__COLOR_CIELuv_Matrix_to_Linear_Rec2020_Matrix",
    "rec2020_to_srgb": "patolette__COLOR_Linear_Rec2020_Matrix_to_sRGB_Matrix",
}


def ref_convert(R, name, flat):
    out = np.array(flat, dtype=np.float64, copy=True)
    m = M2(out.ctypes.data_as(C.POINTER(C.c_double)), out.size // 3, 3)
    f = getattr(R, REF_CONV[name])
    f.argtypes = [C.POINTER(M2)]
    f.restype = None
    f(C.byref(m))
    return out


def color_inputs(n=1500, seed=11):
    img = ob.image(n, seed)
    edge = [0.0, 1.0, 0.04045, 0.0404, 0.0405, 0.5, 1e-9, 0.999999, 0.0031308, 0.003, 0.25, 0.75]
    img[:len(edge)] = edge
    img[n:n + len(edge)] = edge[::-1]
    img[2 * n:2 * n + len(edge)] = edge
    return img


def make_color():
    R = C.CDLL(REF_COLOR)
    n = 1500
    src = color_inputs(n)
    out = {"n": n, "seed": 11}
    ict = ref_convert(R, "srgb_to_ictcp", src)
    luv = ref_convert(R, "srgb_to_cieluv", src)
    rec = ref_convert(R, "srgb_to_rec2020", src)
    out["srgb_to_ictcp"] = ict
    out["srgb_to_cieluv"] = luv
    out["srgb_to_rec2020"] = rec
    out["ictcp_to_rec2020"] = ref_convert(R, "ictcp_to_rec2020", ict)
    out["cieluv_to_rec2020"] = ref_convert(R, "cieluv_to_rec2020", luv)
    out["rec2020_to_srgb"] = ref_convert(R, "rec2020_to_srgb", rec)
    # the CIELuv no-dither chain of patolette.c:305-314, hop by hop through the reference
    chain = ref_convert(R, "srgb_to_ictcp", ref_convert(R, "rec2020_to_srgb", ref_convert(R, "cieluv_to_rec2020", luv)))
    out["cieluv_to_ictcp"] = chain
    np.savez_compressed(os.path.join(OUT, "color_ref.npz"), **out)
    print("color_ref.npz written")


# ----------------------------------------------------------------------------- eigen
def eigen_cases():
    rng = np.random.default_rng(2024)
    cases = []
    for t in range(400):
        k = t % 4
        if k == 0:
            A = np.cov(rng.random((50, 3)).T)
        elif k == 1:
            A = np.cov((rng.random((50, 3)) * [1, 0.01, 0.001]).T)
        elif k == 2:
            A = np.diag(rng.random(3))
        else:
            X = rng.standard_normal((3, 3))
            A = X @ X.T
        cases.append(A)
    # colour-like covariances: ICtCp-scale and Luv-scale
    for t in range(100):
        X = rng.random((200, 3)) * [0.5, 0.1, 0.1]
        cases.append(np.cov(X.T))
        X = rng.random((200, 3)) * [100, 150, 120] - [0, 60, 70]
        cases.append(np.cov(X.T))
    cases += [np.zeros((3, 3)), np.eye(3), np.diag([1, 1, 2.0]), np.diag([2.0, 1, 1]),
              np.array([[2, 1, 0], [1, 2, 0], [0, 0, 1.0]]), np.diag([1e-30, 2e-30, 3e-30]),
              np.array([[1, 0, 1e-3], [0, 1, 0], [1e-3, 0, 1.0]])]
    return np.array(cases)


def make_eigen():
    from scipy.linalg import lapack
    A = eigen_cases()
    W = np.zeros((len(A), 3))
    V = np.zeros((len(A), 3, 3))
    for i, a in enumerate(A):
        w, v, info = lapack.dsyev(np.asfortranarray(a), compute_v=1, lower=1)
        assert info == 0
        W[i] = w
        V[i] = v
    np.savez_compressed(os.path.join(OUT, "eigen_lapack.npz"), A=A, W=W, V=V,
                        note="scipy.linalg.lapack.dsyev, OpenBLAS 0.3.28 (scipy wheel), lower=1")
    print("eigen_lapack.npz written:", len(A), "matrices")


# ----------------------------------------------------------------------------- kmeans
class CP(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("niter nredo verbose spherical int_centroids update_index frozen_centroids "
                                       "min_points_per_centroid max_points_per_centroid seed").split()] + \
               [("decode_block_size", C.c_size_t)]


KM_CASES = [  # n, k, niter, max_samples, weighted, seed, plant_unreachable
    (60000, 256, 1, 256000, False, 1, False),
    (60000, 256, 8, 256000, True, 2, False),
    (100000, 16, 3, 65536, True, 3, False),          # subsampled: 100000 > 16*4096
    (300000, 256, 4, 65536, False, 4, False),        # subsampled to 65536
    (70000, 37, 5, 3700000, True, 5, True),          # k % 8 != 0 + empty clusters -> split_clusters
    (5000, 256, 6, 25600000, False, 6, False),
    (300, 300, 3, 300000, False, 7, False),          # nx == k corner case
    (40000, 13, 32, 65533, True, 8, False),
]


def km_inputs(n, k, weighted, seed, plant):
    x = ob.image(n, seed)                                   # planar f64 in [0,1)
    w = ob.weights(n, seed) if weighted else None
    # initial centres: every (n // k)-th sample (as f32-representable doubles)
    idx = (np.arange(k) * (n // k)).astype(np.int64)
    cent = np.stack([x[idx], x[n + idx], x[2 * n + idx]], axis=1).astype(np.float32).astype(np.float64)
    if plant:
        cent[3] = [9, 9, 9]
        cent[20] = [-5, -5, -5]
    return x, w, cent


def make_kmeans():
    F = C.CDLL(REF_FAISS)
    fp = C.POINTER(C.c_float)
    F.faiss_ClusteringParameters_init.argtypes = [C.POINTER(CP)]
    F.faiss_kmeans_clustering.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, fp, fp, fp, C.POINTER(CP)]
    F.faiss_kmeans_clustering.restype = C.c_int
    out = {"cases": np.array(KM_CASES, dtype=np.int64)}
    for ci, (n, k, niter, max_samples, weighted, seed, plant) in enumerate(KM_CASES):
        x, w, cent = km_inputs(n, k, weighted, seed, plant)
        # what refine.c:102-163 hands to faiss: interleaved f32 samples, f32 weights, f32 centres
        xs = np.ascontiguousarray(x.reshape(3, n).T.astype(np.float32))
        ws = w.astype(np.float32) if w is not None else None
        c = np.ascontiguousarray(cent.astype(np.float32))
        p = CP()
        F.faiss_ClusteringParameters_init(C.byref(p))
        p.niter = niter; p.nredo = 1; p.verbose = 0; p.spherical = 0; p.int_centroids = 0
        p.update_index = 0; p.frozen_centroids = 0; p.min_points_per_centroid = 1
        p.max_points_per_centroid = int(max(max_samples, 256 * 256) // k)      # refine.c:87
        p.seed = 1234; p.decode_block_size = 32768
        rc = F.faiss_kmeans_clustering(3, n, k, xs.ctypes.data_as(fp), c.ctypes.data_as(fp),
                                       ws.ctypes.data_as(fp) if ws is not None else None, C.byref(p))
        out["cent_%d" % ci] = c
        out["rc_%d" % ci] = rc
        # cross-check the oracle right here
        mine = ob.kmeans_refine(x, w, n, cent, niter, max_samples).astype(np.float32)
        print("kmeans case", ci, "rc", rc, "oracle bit-equal:", np.array_equal(mine.view(np.uint32), c.view(np.uint32)))
    np.savez_compressed(os.path.join(OUT, "kmeans_ref.npz"), **out)
    print("kmeans_ref.npz written")


# ----------------------------------------------------------------------------- pipeline (oracle regression)
PIPE_CASES = [  # w, h, K, color_space, niter, dither, weighted, kind, seed
    (256, 256, 16, 2, 0, False, False, "noise", 0),       # BASELINE config 1
    (120, 90, 40, 2, 0, False, False, "noise", 1),
    (200, 150, 64, 1, 0, False, True, "blobs", 2),
    (128, 128, 32, 2, 0, True, False, "blobs", 3),
    (97, 61, 48, 1, 0, True, True, "noise", 4),
    (160, 100, 256, 2, 4, False, False, "noise", 5),
    (300, 200, 24, 2, 0, False, False, "ramp", 6),
    (64, 64, 8, 0, 0, True, False, "blobs", 7),
    (50, 40, 256, 2, 0, False, True, "ramp", 8),
    (300, 300, 256, 2, 8, False, True, "blobs", 9),
    (37, 23, 8, 1, 0, True, False, "noise", 10),
    (640, 480, 256, 2, 0, False, False, "blobs", 11),
    (33, 1, 4, 2, 0, True, False, "noise", 12),
    (16, 16, 300, 2, 0, False, False, "fewcolors", 13),   # fewer distinct colours than K
    (1, 1, 4, 2, 0, False, False, "noise", 14),
]


def pipe_input(w, h, kind, seed, weighted):
    n = w * h
    u = ob.image(n, 100 + seed).reshape(3, n)
    if kind == "noise":
        col = u
    elif kind == "blobs":
        c = ob.image(7, 200 + seed).reshape(3, 7)
        idx = np.minimum((u[0] * 7).astype(np.int64), 6)
        v = ob.image(n, 300 + seed).reshape(3, n)
        col = np.clip(c[:, idx] + 0.1 * (v - 0.5), 0, 1)
    elif kind == "ramp":
        a = np.array([0.1, 0.2, 0.3])[:, None]
        b = np.array([0.9, 0.7, 0.2])[:, None]
        v = ob.image(n, 300 + seed).reshape(3, n)
        col = np.clip(a + u[0][None, :] * (b - a) + 0.004 * (v - 0.5), 0, 1)
    elif kind == "fewcolors":
        c = ob.image(5, 200 + seed).reshape(3, 5)
        idx = np.minimum((u[0] * 5).astype(np.int64), 4)
        col = c[:, idx]
    else:
        raise ValueError(kind)
    flat = np.ascontiguousarray(col).reshape(-1)
    wt = ob.weights(n, seed) if weighted else None
    return flat, wt


def make_pipeline():
    probe = None
    if os.path.exists("/tmp/probe/libpatolette_ref.so"):
        sys.path.insert(0, "/tmp/probe")
        try:
            import run_ref as probe  # noqa
        except Exception:
            probe = None
    out = {"ncases": len(PIPE_CASES)}
    eq = []
    for ci, (w, h, K, cs, niter, dither, weighted, kind, seed) in enumerate(PIPE_CASES):
        flat, wt = pipe_input(w, h, kind, seed, weighted)
        ec, pal, pmap = ob.patolette(w, h, flat, wt, K, dither=dither, color_space=cs, kmeans_niter=niter,
                                     kmeans_max_samples=65536)
        out["ec_%d" % ci] = ec
        out["pal_%d" % ci] = np.asarray(pal)
        out["map_%d" % ci] = pmap.astype(np.uint16)
        ok = -1
        if probe is not None:
            n = w * h
            col = flat.reshape(3, n).T.copy()
            ec2, pal2, pmap2, _ = probe.quantize(w, h, col, K, dither=dither, cs=cs, niter=niter, weights=wt,
                                                 max_samples=65536)
            if w * h == 1 and dither is False:
                pass
            ok = int(ec2 == ec and np.array_equal(pal2, np.asarray(pal), equal_nan=True) and np.array_equal(pmap2, pmap))
        eq.append(ok)
        print("pipeline case", ci, (w, h, K, cs, niter, dither, weighted, kind), "ec", ec, "probe_equal", ok)
    out["probe_equal"] = np.array(eq)
    np.savez_compressed(os.path.join(OUT, "pipeline_oracle.npz"), **out)
    print("pipeline_oracle.npz written")


if __name__ == "__main__":
    which = sys.argv[1:] or ["color", "eigen", "kmeans", "pipeline"]
    if "color" in which:
        make_color()
    if "eigen" in which:
        make_eigen()
    if "kmeans" in which:
        make_kmeans()
    if "pipeline" in which:
        make_pipeline()
