"""ctypes binding of the CPU oracle (oracle/patolette_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package `patolette_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")


def build(force=False):
    """Compile the oracle (and, where /root/reference exists, oracle/_ref) via the Makefile."""
    src = os.path.join(HERE, "patolette_oracle.c")
    stale = (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    return LIB_PATH


class Options(C.Structure):
    # lib/include/patolette.h:13-20
    _fields_ = [("dither", C.c_bool), ("palette_only", C.c_bool), ("color_space", C.c_int),
                ("kmeans_niter", C.c_int), ("kmeans_max_samples", C.c_size_t), ("verbose", C.c_bool)]


class SplitRecord(C.Structure):
    # orc_SplitRecord (patolette_oracle.h) == patolette_amd__SplitRecord (include/patolette_amd.h)
    _fields_ = [("row", C.c_int32), ("new_row", C.c_int32), ("split", C.c_int32), ("degenerate", C.c_int32),
                ("n", C.c_uint64), ("n_left", C.c_uint64), ("n_right", C.c_uint64), ("sw", C.c_double),
                ("axis", C.c_double * 3), ("cov6", C.c_double * 6), ("dist", C.c_double), ("dist_left", C.c_double),
                ("dist_right", C.c_double), ("benefit", C.c_double)]


class SplitTraceHeader(C.Structure):
    _fields_ = [("n_base", C.c_int32), ("n_clusters", C.c_int32), ("n_records", C.c_int32), ("stopped_early", C.c_int32),
                ("gq_axis", C.c_double * 3), ("gq_cuts", C.c_uint64 * 14), ("gq_cov6", C.c_double * 6)]


def trace_to_dict(hdr, recs):
    """ctypes header + records -> plain Python (the form tests/tie_prover.py works on)."""
    return dict(n_base=hdr.n_base, n_clusters=hdr.n_clusters, stopped_early=bool(hdr.stopped_early),
                gq_axis=[float(v) for v in hdr.gq_axis], gq_cov6=[float(v) for v in hdr.gq_cov6], gq_cuts=[int(v) for v in hdr.gq_cuts][:hdr.n_base + 1],
                splits=[dict(row=r.row, new_row=r.new_row, split=r.split, degenerate=r.degenerate, n=r.n, n_left=r.n_left,
                             n_right=r.n_right, sw=r.sw, axis=[float(v) for v in r.axis], cov6=[float(v) for v in r.cov6],
                             dist=r.dist, dist_left=r.dist_left, dist_right=r.dist_right, benefit=r.benefit) for r in recs])


_lib = None
dp = C.POINTER(C.c_double)
fp = C.POINTER(C.c_float)
zp = C.POINTER(C.c_size_t)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_splitmix64.argtypes = [C.c_uint64]
        for name in ("orc_fill_uniform", "orc_fill_image", "orc_fill_weights"):
            getattr(L, name).argtypes = [dp, C.c_size_t, C.c_uint64]
            getattr(L, name).restype = None
        for name in ("orc_srgb_to_ictcp", "orc_srgb_to_cieluv", "orc_ictcp_to_rec2020",
                     "orc_cieluv_to_rec2020", "orc_srgb_to_rec2020", "orc_rec2020_to_srgb"):
            getattr(L, name).argtypes = [dp, C.c_size_t]
            getattr(L, name).restype = None
        L.orc_eigen_sym3.argtypes = [dp, dp]
        L.orc_eigen_sym3.restype = C.c_int
        L.orc_pca_axis.argtypes = [dp, dp, C.c_size_t, dp, dp]
        L.orc_pca_axis.restype = C.c_int
        L.orc_axis_sort.argtypes = [dp, C.c_size_t, dp, C.c_size_t, zp]
        L.orc_axis_sort.restype = None
        L.orc_quantize_clusters.argtypes = [dp, dp, C.c_size_t, C.c_size_t, dp, zp,
                                            C.POINTER(C.c_uint32), zp, zp, zp]
        L.orc_quantize_clusters.restype = C.c_int
        L.orc_kmeans_refine.argtypes = [dp, dp, C.c_size_t, dp, C.c_size_t, C.c_int, C.c_size_t]
        L.orc_kmeans_refine.restype = None
        L.orc_kmeans_subsample_indices.argtypes = [C.c_size_t, C.c_size_t, C.c_int64, C.POINTER(C.c_int32)]
        L.orc_kmeans_subsample_indices.restype = None
        L.orc_kmeans_assign.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.POINTER(C.c_int64), fp]
        L.orc_kmeans_assign.restype = None
        L.orc_kmeans_update.argtypes = [fp, fp, C.c_size_t, C.POINTER(C.c_int64), fp, C.c_size_t, fp]
        L.orc_kmeans_update.restype = None
        L.orc_kmeans_split_clusters.argtypes = [C.c_size_t, C.c_size_t, fp, fp]
        L.orc_kmeans_split_clusters.restype = C.c_int
        L.orc_nn_map.argtypes = [dp, C.c_size_t, dp, C.c_size_t, zp]
        L.orc_nn_map.restype = None
        L.orc_dither_riemersma.argtypes = [dp, C.c_size_t, C.c_size_t, dp, C.c_size_t, zp]
        L.orc_dither_riemersma.restype = None
        L.orc_dither_riemersma_prefix.argtypes = [dp, C.c_size_t, C.c_size_t, dp, C.c_size_t, zp, C.c_size_t]
        L.orc_dither_riemersma_prefix.restype = None
        L.orc_hilbert_order.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64)]
        L.orc_hilbert_order.restype = C.c_size_t
        L.orc_mbd.argtypes = [C.c_size_t, C.c_size_t, fp, C.c_int, fp]
        L.orc_mbd.restype = C.c_int
        L.orc_patolette.argtypes = [C.c_size_t, C.c_size_t, dp, dp, C.c_size_t, C.POINTER(Options),
                                    dp, zp, C.POINTER(C.c_int)]
        L.orc_patolette.restype = None
        L.orc_patolette_from_centers.argtypes = [C.c_size_t, C.c_size_t, dp, dp, C.c_size_t, C.POINTER(Options), dp, C.c_size_t, dp, zp]
        L.orc_patolette_from_centers.restype = None
        L.orc_exit_message.argtypes = [C.c_int]
        L.orc_exit_message.restype = C.c_char_p
        L.orc_last_split_trace.argtypes = [C.POINTER(SplitTraceHeader), C.POINTER(SplitRecord), C.c_size_t]
        L.orc_last_split_trace.restype = C.c_size_t
        L.orc_set_sum_reversed.argtypes = [C.c_int]
        L.orc_set_sum_reversed.restype = None
        L.orc_set_fault.argtypes = [C.c_int]
        L.orc_set_fault.restype = None
        L.orc_last_timings.argtypes = [dp]
        L.orc_last_timings.restype = None
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads.restype = None
        L.orc_get_threads.restype = C.c_int
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(dp) if a is not None else None


def planar(a):
    """(N,3) array -> contiguous planar copy (column-major (N,3)) as a flat f64 array of 3N."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).T).reshape(-1)


def unplanar(flat, n):
    return np.asarray(flat).reshape(3, n).T.copy()


def image(n, seed=0):
    out = np.empty(3 * n, dtype=np.float64)
    lib().orc_fill_image(_d(out), n, seed)
    return out


def weights(n, seed=0):
    out = np.empty(n, dtype=np.float64)
    lib().orc_fill_weights(_d(out), n, seed)
    return out


def convert(name, flat):
    """name in {srgb_to_ictcp, srgb_to_cieluv, ictcp_to_rec2020, cieluv_to_rec2020,
    srgb_to_rec2020, rec2020_to_srgb}; returns a converted copy of the planar array."""
    out = np.array(flat, dtype=np.float64, copy=True)
    getattr(lib(), "orc_" + name)(_d(out), out.size // 3)
    return out


def eigen_sym3(a):
    """a: (3,3) symmetric (lower triangle read). Returns (w ascending, V columns)."""
    buf = np.asfortranarray(np.array(a, dtype=np.float64)).reshape(-1, order="F").copy()
    w = np.zeros(3)
    info = lib().orc_eigen_sym3(_d(buf), _d(w))
    return info, w, buf.reshape(3, 3, order="F")


def quantize_clusters(flat, w, n, K, want_membership=True):
    centers = np.full(3 * K, np.nan)
    ncl = C.c_size_t(0)
    nbase = C.c_size_t(0)
    sev = C.c_size_t(0)
    spx = C.c_size_t(0)
    member = np.zeros(n, dtype=np.uint32) if want_membership else None
    rc = lib().orc_quantize_clusters(_d(flat), _d(w), n, K, _d(centers), C.byref(ncl),
                                     member.ctypes.data_as(C.POINTER(C.c_uint32)) if member is not None else None,
                                     C.byref(nbase), C.byref(sev), C.byref(spx))
    return dict(rc=rc, centers=centers.reshape(3, K).T.copy(), n_clusters=ncl.value, member=member,
                n_base=nbase.value, split_evals=sev.value, split_px=spx.value)


def last_split_trace():
    """Split trace of the last quantize_clusters / patolette call (orc_last_split_trace) as a dict."""
    hdr = SplitTraceHeader()
    n = lib().orc_last_split_trace(C.byref(hdr), None, 0)
    recs = (SplitRecord * max(1, n))()
    lib().orc_last_split_trace(C.byref(hdr), recs, n)
    return trace_to_dict(hdr, recs[:n])


def set_sum_reversed(on):
    """Test knob: the reference's sequential sums taken back to front (another member of its rounding-noise set)."""
    lib().orc_set_sum_reversed(int(bool(on)))


def set_fault(which):
    """Test knob: 0 none; 1 wrong cut; 2 wrong greedy step; 3 last arg-max (a tie-set member)."""
    lib().orc_set_fault(int(which))


def kmeans_refine(flat, w, n, centers, niter, max_samples):
    """centers: (k,3). Returns refined (k,3)."""
    k = centers.shape[0]
    c = planar(centers).copy()
    lib().orc_kmeans_refine(_d(flat), _d(w), n, _d(c), k, niter, max_samples)
    return c.reshape(3, k).T.copy()


def nn_map(flat, n, palette):
    k = palette.shape[0]
    p = planar(palette)
    out = np.zeros(n, dtype=np.uintp)
    lib().orc_nn_map(_d(flat), n, _d(p), k, out.ctypes.data_as(zp))
    return out


def dither(flat, width, height, palette, init=None):
    k = palette.shape[0]
    p = planar(palette)
    out = np.zeros(width * height, dtype=np.uintp) if init is None else init.astype(np.uintp).copy()
    lib().orc_dither_riemersma(_d(flat), width, height, _d(p), k, out.ctypes.data_as(zp))
    return out


def dither_prefix(flat, width, height, palette, max_visits, fill=0xFFFF):
    k = palette.shape[0]
    p = planar(palette)
    out = np.full(width * height, fill, dtype=np.uintp)
    lib().orc_dither_riemersma_prefix(_d(flat), width, height, _d(p), k, out.ctypes.data_as(zp), max_visits)
    return out


def hilbert_order(width, height):
    out = np.zeros(max(1, width * height), dtype=np.uint64)
    n = lib().orc_hilbert_order(width, height, out.ctypes.data_as(C.POINTER(C.c_uint64)))
    return out[:n].copy()


def mbd(img32, iters=3):
    """img32: (rows, cols) float32.  Returns D (rows, cols) float32, or None for images <= 3 in a dimension."""
    img32 = np.ascontiguousarray(img32, dtype=np.float32)
    rows, cols = img32.shape
    out = np.zeros((rows, cols), dtype=np.float32)
    rc = lib().orc_mbd(rows, cols, img32.ctypes.data_as(fp), iters, out.ctypes.data_as(fp))
    return None if rc != 0 else out


def patolette(width, height, flat, w, K, dither=True, palette_only=False, color_space=2,
              kmeans_niter=32, kmeans_max_samples=512 ** 2):
    """Mirror of the C patolette(): returns (exit_code, palette (K,3) F-order, palette_map or None)."""
    opt = Options(dither, palette_only, color_space, kmeans_niter, kmeans_max_samples, False)
    pal = np.zeros((K, 3), dtype=np.float64, order="F")
    n = width * height
    pmap = None if palette_only else np.zeros(n, dtype=np.uintp)
    code = C.c_int(0)
    lib().orc_patolette(width, height, _d(flat), _d(w), K, C.byref(opt),
                        pal.ctypes.data_as(dp) if K > 0 else None,
                        pmap.ctypes.data_as(zp) if pmap is not None and n > 0 else None, C.byref(code))
    return code.value, pal, pmap


def patolette_from_centers(width, height, flat, w, K, centers, dither=True, palette_only=False, color_space=2,
                           kmeans_niter=32, kmeans_max_samples=512 ** 2):
    """The reference's path behind the local quantiser (patolette.c:246-336) from given cluster centres ((len,3), quantisation
    space): returns (palette (K,3) F-order, palette_map or None)."""
    opt = Options(dither, palette_only, color_space, kmeans_niter, kmeans_max_samples, False)
    pal = np.zeros((K, 3), dtype=np.float64, order="F")
    n = width * height
    pmap = None if palette_only else np.zeros(n, dtype=np.uintp)
    cen = planar(centers)
    lib().orc_patolette_from_centers(width, height, _d(flat), _d(w), K, C.byref(opt), _d(cen), len(centers),
                                     pal.ctypes.data_as(dp), pmap.ctypes.data_as(zp) if pmap is not None else None)
    return pal, pmap


def set_threads(n):
    """Threads for the loops the reference's dependencies thread (faiss assign / update, FLANN search); results do not depend on it."""
    lib().orc_set_threads(int(n))


def last_timings():
    t = np.zeros(6)
    lib().orc_last_timings(_d(t))
    return dict(zip(("convert", "gq", "lq", "kmeans", "map", "total"), t.tolist()))
