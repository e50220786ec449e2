"""ctypes binding of libpatolette_amd.so -- the C-ABI boundary of the MI355X-native path.

There is no CPU fallback: if the shared library (or a HIP device, at call time) is missing,
everything here raises.  Build the library with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C patolette_amd/csrc`.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libpatolette_amd.so")

dp = C.POINTER(C.c_double)
zp = C.POINTER(C.c_size_t)


class QuantizationOptions(C.Structure):
    """patolette__QuantizationOptions, reference lib/include/patolette.h:13-20."""
    _fields_ = [("dither", C.c_bool), ("palette_only", C.c_bool), ("color_space", C.c_int),
                ("kmeans_niter", C.c_int), ("kmeans_max_samples", C.c_size_t), ("verbose", C.c_bool)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ms_total", "ms_upload", "ms_convert", "ms_gq", "ms_lq",
                                          "ms_kmeans", "ms_map", "ms_download", "ms_saliency")] + \
               [(n, C.c_size_t) for n in ("n_base_clusters", "n_clusters", "split_evals", "split_px",
                                          "lq_rounds", "kmeans_samples", "dither_segments", "dither_repairs",
                                          "dither_rounds", "dither_through", "dither_jumps", "dither_solo")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SplitRecord(C.Structure):
    """patolette_amd__SplitRecord (include/patolette_amd.h): one committed split of the greedy loop, local.c:347-390."""
    _fields_ = [("row", C.c_int32), ("new_row", C.c_int32), ("split", C.c_int32), ("degenerate", C.c_int32),
                ("n", C.c_uint64), ("n_left", C.c_uint64), ("n_right", C.c_uint64), ("sw", C.c_double),
                ("axis", C.c_double * 3), ("cov6", C.c_double * 6), ("dist", C.c_double), ("dist_left", C.c_double),
                ("dist_right", C.c_double), ("benefit", C.c_double)]


class SplitTrace(C.Structure):
    """patolette_amd__SplitTrace: what the global quantiser decided (global.c:388-443) + counts."""
    _fields_ = [("n_base", C.c_int32), ("n_clusters", C.c_int32), ("n_records", C.c_int32), ("stopped_early", C.c_int32),
                ("gq_axis", C.c_double * 3), ("gq_cuts", C.c_uint64 * 14), ("gq_cov6", C.c_double * 6)]


ALLREDUCE_SUM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)


class Comm(C.Structure):
    """patolette_amd__Comm: the element-wise in-place SUM a group of GPUs lends to patolette_amd_slice."""
    _fields_ = [("rank", C.c_int), ("size", C.c_int), ("allreduce_sum", ALLREDUCE_SUM_FN), ("ctx", C.c_void_p),
                ("host_buffers", C.c_int)]


# every symbol include/patolette.h and include/patolette_amd.h declare
SYMBOLS = {
    # --- include/patolette.h (the reference's ABI)
    "patolette": (None, [C.c_size_t, C.c_size_t, dp, dp, C.c_size_t, C.POINTER(QuantizationOptions), dp, zp,
                         C.POINTER(C.c_int)]),
    "get_patolette_exit_code_info_message": (C.c_char_p, [C.c_int]),
    "patolette_create_default_options": (C.POINTER(QuantizationOptions), []),
    # --- include/patolette_amd.h (additive)
    "patolette_amd_device_count": (C.c_int, []),
    "patolette_amd_set_device": (C.c_int, [C.c_int]),
    "patolette_amd_last_error": (C.c_char_p, []),
    "patolette_amd_malloc": (C.c_void_p, [C.c_size_t]),
    "patolette_amd_free": (None, [C.c_void_p]),
    "patolette_amd_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "patolette_amd_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "patolette_amd_synchronize": (C.c_int, []),
    "patolette_amd_release_workspace": (None, []),
    "patolette_amd_batch_dmap": (None, [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p), C.c_int, C.POINTER(dp), C.c_double,
                                        C.c_size_t, C.POINTER(QuantizationOptions), C.POINTER(dp), C.c_void_p, C.c_int,
                                        C.POINTER(C.c_int)]),
    "patolette_amd_slice": (None, [C.c_size_t, C.c_size_t, C.c_size_t, dp, dp, C.c_size_t, C.POINTER(QuantizationOptions),
                                   C.POINTER(Comm), dp, zp, C.POINTER(C.c_int)]),
    "patolette_amd_slice_device": (None, [C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.POINTER(QuantizationOptions), C.POINTER(Comm), dp, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "patolette_amd_set_invariant_sums": (C.c_int, [C.c_int]),
    "patolette_amd_set_kmeans_update": (C.c_int, [C.c_int]),
    "patolette_amd_set_subsample_cache": (C.c_int, [C.c_int]),
    "patolette_amd_subsample_indices": (C.c_int, [C.c_size_t, C.c_size_t, C.POINTER(C.c_int32)]),
    "patolette_amd_eigen_sym3": (C.c_int, [dp, dp, dp]),
    "patolette_amd_principal_axis": (C.c_int, [dp, dp]),
    "patolette_amd_eigen_sym3_device": (C.c_int, [dp, C.c_size_t, dp, dp, C.POINTER(C.c_int)]),
    "patolette_amd_fill_image": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64]),
    "patolette_amd_fill_weights": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64]),
    "patolette_amd_device": (None, [C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.POINTER(QuantizationOptions), dp, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "patolette_amd_quantize": (None, [C.c_size_t, C.c_size_t, dp, dp, C.c_double, C.c_size_t, C.POINTER(QuantizationOptions), dp, zp,
                                      C.POINTER(C.c_int)]),
    "patolette_amd_quantize_rows": (None, [C.c_size_t, C.c_size_t, dp, dp, C.c_double, C.c_size_t, C.POINTER(QuantizationOptions), dp,
                                           zp, C.POINTER(C.c_int)]),
    "patolette_amd_saliency_weights": (C.c_int, [C.c_size_t, C.c_size_t, dp, C.c_double, dp]),
    "patolette_amd_mbd": (C.c_int, [C.c_size_t, C.c_size_t, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]),
    "patolette_amd_u8": (None, [C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, dp, C.c_double, C.c_size_t,
                                C.POINTER(QuantizationOptions), dp, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                C.POINTER(C.c_int)]),
    "patolette_amd_u8_device": (None, [C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_size_t,
                                       C.POINTER(QuantizationOptions), dp, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.POINTER(C.c_int)]),
    "patolette_amd_batch": (None, [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(dp), C.POINTER(dp), C.c_double, C.c_size_t,
                                   C.POINTER(QuantizationOptions), C.POINTER(dp), C.POINTER(zp), C.POINTER(C.c_int)]),
    "patolette_amd_batch_rows": (None, [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(dp), C.POINTER(dp), C.c_double, C.c_size_t,
                                        C.POINTER(QuantizationOptions), C.POINTER(dp), C.POINTER(zp), C.POINTER(C.c_int)]),
    "patolette_amd_batch_u8": (None, [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p), C.c_int, C.POINTER(dp), C.c_double, C.c_size_t,
                                      C.POINTER(QuantizationOptions), C.POINTER(dp), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "patolette_amd_pow": (C.c_int, [dp, C.c_double, dp, C.c_size_t]),
    "patolette_amd_convert": (C.c_int, [C.c_int, dp, C.c_size_t]),
    "patolette_amd_quantize_clusters": (C.c_int, [dp, dp, C.c_size_t, C.c_size_t, dp, zp]),
    "patolette_amd_kmeans_refine": (C.c_int, [dp, dp, C.c_size_t, dp, C.c_size_t, C.c_int, C.c_size_t]),
    "patolette_amd_nn_map": (C.c_int, [dp, C.c_size_t, dp, C.c_size_t, zp]),
    "patolette_amd_set_global_quantiser": (C.c_int, [C.c_int]),
    "patolette_amd_dither": (C.c_int, [dp, C.c_size_t, C.c_size_t, dp, C.c_size_t, zp]),
    "patolette_amd_dither_config": (None, [C.c_int, C.c_int]),
    "patolette_amd_dither_layout": (None, [C.c_int]),
    "patolette_amd_debug_dither_solo_cap": (C.c_int, [C.c_int]),
    "patolette_amd_debug_dither_stall_passes": (C.c_int, [C.c_int]),
    "patolette_amd_dither_layout_in_use": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t]),
    "patolette_amd_debug_dither_locate": (None, [C.c_size_t, C.c_size_t, C.c_ulonglong, C.POINTER(C.c_ulonglong),
                                                 C.POINTER(C.c_ulonglong)]),
    "patolette_amd_last_stats": (None, [C.POINTER(Stats)]),
    "patolette_amd_last_map_palette": (C.c_size_t, [dp, C.c_size_t]),
    "patolette_amd_last_split_trace": (C.c_size_t, [C.POINTER(SplitTrace), C.POINTER(SplitRecord), C.c_size_t]),
    "patolette_amd_last_cluster_centers": (C.c_size_t, [dp, C.c_size_t]),
    "patolette_amd_debug_fault": (C.c_int, [C.c_int]),
    "patolette_amd_set_split_loop": (C.c_int, [C.c_int]),
    "patolette_amd_profile_enable": (None, [C.c_int]),
    "patolette_amd_profile_only": (None, [C.c_char_p]),
    "patolette_amd_profile_sample": (None, [C.c_int]),
    "patolette_amd_profile_count": (C.c_int, []),
    "patolette_amd_profile_get": (C.c_int, [C.c_int, C.c_char_p, dp, zp, dp]),
}

_lib = None


def lib():
    """Load libpatolette_amd.so (raises if it has not been built) and type its entry points."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "patolette_amd: %s not found -- build it with `make -C patolette_amd/csrc` "
                "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)          # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().patolette_amd_last_error().decode("utf-8", "replace")


def last_stats():
    s = Stats()
    lib().patolette_amd_last_stats(C.byref(s))
    return s.as_dict()


def last_split_trace():
    """The quantisers' decisions of the last call on this thread (patolette_amd_last_split_trace) as a plain dict:
    n_base, n_clusters, stopped_early, gq_axis, gq_cov6, gq_cuts, splits[...]."""
    hdr = SplitTrace()
    n = lib().patolette_amd_last_split_trace(C.byref(hdr), None, 0)
    recs = (SplitRecord * max(1, n))()
    lib().patolette_amd_last_split_trace(C.byref(hdr), recs, n)
    return dict(n_base=hdr.n_base, n_clusters=hdr.n_clusters, stopped_early=bool(hdr.stopped_early),
                gq_axis=[float(v) for v in hdr.gq_axis], gq_cov6=[float(v) for v in hdr.gq_cov6],
                gq_cuts=[int(v) for v in hdr.gq_cuts][:hdr.n_base + 1],
                splits=[dict(row=r.row, new_row=r.new_row, split=r.split, degenerate=r.degenerate, n=r.n, n_left=r.n_left,
                             n_right=r.n_right, sw=r.sw, axis=[float(v) for v in r.axis], cov6=[float(v) for v in r.cov6],
                             dist=r.dist, dist_left=r.dist_left, dist_right=r.dist_right, benefit=r.benefit) for r in recs[:n]])


def last_cluster_centers():
    """PALETTE_create's rows of the last call, (len, 3), in the quantisation space (before any KMeans)."""
    import numpy as np
    n = lib().patolette_amd_last_cluster_centers(None, 0)
    out = np.zeros((max(1, n), 3), order="F")
    lib().patolette_amd_last_cluster_centers(out.ctypes.data_as(dp), max(1, n))
    return np.ascontiguousarray(out[:n])


def profile(enable=True, only=None, sample=1):
    """Per-kernel HIP-event timing on/off (resets the counters); `only` restricts it to one kernel name, `sample` to every
    sample-th launch of that kernel."""
    lib().patolette_amd_profile_only(only.encode() if only else None)
    lib().patolette_amd_profile_sample(int(sample) if only else 1)
    lib().patolette_amd_profile_enable(1 if enable else 0)


def profile_results():
    L = lib()
    out = {}
    for i in range(L.patolette_amd_profile_count()):
        name = C.create_string_buffer(64)
        ms = C.c_double(0)
        n = C.c_size_t(0)
        by = C.c_double(0)
        if L.patolette_amd_profile_get(i, name, C.byref(ms), C.byref(n), C.byref(by)) == 0:
            out[name.value.decode()] = {"total_ms": ms.value, "launches": n.value, "bytes": by.value}
    return out
