"""The C-ABI boundary without a GPU: the library loads, and exports every symbol that
include/patolette.h and include/patolette_amd.h declare; struct layout matches the reference."""
import ctypes as C
import os
import re

from tests.util import ROOT


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", txt)
    return {n for n in names if n.startswith("patolette") or n.startswith("get_patolette")}


def test_every_declared_symbol_is_exported(native):
    L = native.lib()
    declared = declared_functions("patolette.h") | declared_functions("patolette_amd.h")
    assert {"patolette", "get_patolette_exit_code_info_message", "patolette_create_default_options"} <= declared
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(native.SYMBOLS), declared ^ set(native.SYMBOLS)


def test_options_struct_layout_matches_reference(native):
    # lib/include/patolette.h:13-20 on x86-64 SysV: offsets 0,1,4,8,16,24, sizeof 32
    Q = native.QuantizationOptions
    assert [getattr(Q, n).offset for n, _ in Q._fields_] == [0, 1, 4, 8, 16, 24]
    assert C.sizeof(Q) == 32


def test_default_options_and_messages(native):
    L = native.lib()
    o = L.patolette_create_default_options()                    # patolette.c:107-119
    assert (o.contents.dither, o.contents.palette_only, o.contents.color_space, o.contents.kmeans_niter,
            o.contents.kmeans_max_samples, o.contents.verbose) == (True, False, 2, 32, 512 * 512, False)
    C.CDLL(None).free(C.cast(o, C.c_void_p))
    msgs = [L.get_patolette_exit_code_info_message(-i).decode() for i in range(5)]   # patolette.c:32-38
    assert msgs == ["Quantization successful.", "Internal quantization error.", "Image dimensions should be greater than 0.",
                    "Palette size should be greater than 0.", "Image dimensions are too big."]


def test_argument_validation_needs_no_gpu(native):
    # patolette.c:61-95 runs before anything touches the device
    L = native.lib()
    opts = native.QuantizationOptions(False, True, 2, 0, 0, False)
    code = C.c_int(7)
    L.patolette(0, 5, None, None, 4, C.byref(opts), None, None, C.byref(code))
    assert code.value == -2
    L.patolette(3, 5, None, None, 0, C.byref(opts), None, None, C.byref(code))
    assert code.value == -3
    L.patolette(50000, 50000, None, None, 4, C.byref(opts), None, None, C.byref(code))
    assert code.value == -4


def test_python_surface_validation_messages():
    # patolette.pyx:328-373: returned before crossing into C
    import numpy as np
    import patolette_amd as p
    assert p.quantize(2, 2, np.zeros((4, 4)), 2) == (False, None, None,
                                                     "Expected colors to be in sRGB[0, 1] space. Channel count mismatch: 4 found.")
    assert p.quantize(2, 3, np.zeros((4, 3)), 2) == (False, None, None,
                                                     "The number of colors doesn't match the supplied width and height.")
    assert p.quantize(2, 2, np.zeros((4, 3)), 2, tile_size=-1) == (False, None, None,
                                                                    "tile_size parameter expected to be in the range [0, inf]")
    assert (p.ColorSpace_sRGB, p.ColorSpace_CIELuv, p.ColorSpace_ICtCp) == (0, 1, 2)
    assert {"quantize", "ColorSpace_sRGB", "ColorSpace_CIELuv", "ColorSpace_ICtCp"} <= set(p.__all__)


def test_batch_entry_rejects_malformed_weights(native):
    """quantize_batch validates the weights before anything reaches the C ABI (a short array would be read out of bounds)."""
    import numpy as np
    import pytest
    import patolette_amd as p
    imgs = [np.zeros((12, 3)), np.zeros((12, 3))]
    with pytest.raises(ValueError):
        p.quantize_batch(4, 3, imgs, 2, weights=[np.ones(12)])              # one entry per image
    with pytest.raises(ValueError):
        p.quantize_batch(4, 3, imgs, 2, weights=[np.ones(12), np.ones(5)])  # width*height values each
    u8 = [np.zeros((3, 4, 3), dtype=np.uint8)] * 2
    with pytest.raises(ValueError):
        p.quantize_u8_batch(u8, 2, weights=[np.ones(12)])
    from patolette_amd import dist as pdist
    with pytest.raises(ValueError):
        pdist.quantize_batch_sharded(4, 3, imgs, 2, weights=[np.ones(12), np.ones(11)], quantize_fn=lambda *a, **k: None)


def test_slice_entry_validation_needs_no_gpu(native):
    """patolette_amd_slice rejects malformed calls before touching a device or the caller's collective."""
    import numpy as np
    from patolette_amd import _native
    calls = []
    cb = _native.ALLREDUCE_SUM_FN(lambda ctx, buf, count, dtype: calls.append(1) or 0)
    comm = _native.Comm(0, 2, cb, None, 1)
    opts = _native.QuantizationOptions(False, False, 2, 0, 512 ** 2, False)
    px = np.zeros(3 * 10)
    pal = np.zeros(3 * 4)
    mp = np.zeros(10, dtype=np.uintp)

    def call(total, begin, count, K, comm_ref, data=px, pmap=mp):
        code = C.c_int(99)
        native.lib().patolette_amd_slice(total, begin, count, data.ctypes.data_as(_native.dp) if data is not None else None, None, K,
                                   C.byref(opts), comm_ref, pal.ctypes.data_as(_native.dp),
                                   pmap.ctypes.data_as(_native.zp) if pmap is not None else None, C.byref(code))
        return code.value
    assert call(0, 0, 10, 4, C.byref(comm)) == -2                     # no pixels
    assert call(20, 0, 10, 0, C.byref(comm)) == -3                    # no colours
    assert call(20, 0, 10, 4, None) == -1                             # no group
    assert call(20, 15, 10, 4, C.byref(comm)) == -1                   # slice beyond the image
    assert call(20, 0, 0, 4, C.byref(comm)) == -1                     # empty slice
    assert call(20, 0, 10, 4, C.byref(comm), pmap=None) == -1         # a map is wanted but there is nowhere to put it
    bad = _native.Comm(2, 2, cb, None, 1)
    assert call(20, 0, 10, 4, C.byref(bad)) == -1                     # rank outside the group
    assert not calls


def test_product_subsample_list_is_the_rand_perm_prefix_of_faiss(native, ob):
    """The product's host code for the KMeans subsample (two-pass sparse Fisher-Yates with prefetched table slots, host_math.h)
    against numpy's MT19937 driving a full Fisher-Yates (random.cpp:184-194) and against the oracle's list at the sizes the
    pipeline uses (262 144 of 16.8 M; n barely above take; take == n; tiny n)."""
    import numpy as np
    L = native.lib()
    i32p = C.POINTER(C.c_int32)
    n, take = 5000, 5000
    mt = np.random.MT19937()
    mt._legacy_seeding(1234)
    raw = mt.random_raw(n)
    perm = np.arange(n)
    for i in range(n - 1):
        j = i + int(raw[i]) % (n - i)
        perm[i], perm[j] = perm[j], perm[i]
    for tk in (1, 700, 4999, 5000):
        out = np.zeros(tk, dtype=np.int32)
        assert L.patolette_amd_subsample_indices(n, tk, out.ctypes.data_as(i32p)) == 0
        assert np.array_equal(out, perm[:tk]), tk
    for n, tk in ((4096 * 4096, 262144), (262145, 262144), (300000, 262144), (70000, 65536), (1, 1), (2, 2)):
        a = np.zeros(tk, dtype=np.int32)
        b = np.zeros(tk, dtype=np.int32)
        assert L.patolette_amd_subsample_indices(n, tk, a.ctypes.data_as(i32p)) == 0
        ob.lib().orc_kmeans_subsample_indices(n, tk, 1234, b.ctypes.data_as(i32p))
        assert np.array_equal(a, b), (n, tk)
        assert len(np.unique(a)) == tk and a.min() >= 0 and a.max() < n
