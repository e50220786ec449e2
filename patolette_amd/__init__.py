"""patolette_amd -- MI355X-native drop-in for big-nacho/patolette's `patolette.quantize()`.

Mirrors the reference's Python surface (src/patolette/patolette.pyx:324-344, 441-473):
`quantize(...)` with the same positional/keyword arguments, the same validation messages and
the same `(success, palette, palette_map, message)` return tuple, plus the `ColorSpace_*`
constants.  The work behind it runs in hand-written HIP kernels on gfx950 through the C ABI of
`libpatolette_amd.so` (include/patolette.h); there is no CPU fallback.

`tile_size > 0` derives the saliency weights on the GPU as the reference's binding does on the CPU
(patolette.pyx:203-313).  Additive (not in the reference): the `weights=` keyword (explicit
per-pixel weights instead of the saliency-derived ones), `saliency_weights`, `quantize_batch`, the 8-bit adaptor `quantize_u8` and its
batch form `quantize_u8_batch`.
"""
import ctypes as C

import numpy as np

from . import _native
from ._native import last_stats, profile, profile_results  # noqa: F401

__version__ = "0.1.0"

# patolette.pyx:324-326
ColorSpace_sRGB = 0
ColorSpace_CIELuv = 1
ColorSpace_ICtCp = 2

# patolette.pyx:328-330
color_mismatch = "The number of colors doesn't match the supplied width and height."
bad_channel_count = 'Expected colors to be in sRGB[0, 1] space. Channel count mismatch: {} found.'
bad_tile_size = 'tile_size parameter expected to be in the range [0, inf]'



def _raise_saliency(code, message):
    """Exit codes of the saliency stage: the reference raises Python exceptions in the same situations
    (patolette.pyx:157-158 / :228-232 shape, numpy LinAlgError for a singular border covariance)."""
    if code == -5:
        raise ValueError(message)
    if code == -6:
        raise np.linalg.LinAlgError("Singular matrix (%s)" % message)


def _dp(a):
    return a.ctypes.data_as(_native.dp) if a is not None and a.size > 0 else None


def quantize(width, height, colors, palette_size, dither=True, palette_only=False,
             color_space=ColorSpace_ICtCp, tile_size=512, kmeans_niter=32, kmeans_max_samples=512 ** 2,
             verbose=False, weights=None):
    """Quantise an image; reference `patolette.quantize` (patolette.pyx:332-466).

    colors: (width*height, 3) float64, sRGB in [0,1], row-scan pixel order.
    Returns (success, palette (K,3) F-ordered | None, palette_map (N,) uintp | None, message).
    """
    colors = np.asarray(colors)
    if colors.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % colors.ndim)
    color_count, channel_count = colors.shape
    # validations the reference does before crossing into C (patolette.pyx:351-373)
    if channel_count != 3:
        return (False, None, None, bad_channel_count.format(channel_count))
    if color_count != width * height:
        return (False, None, None, color_mismatch)
    if tile_size < 0:
        return (False, None, None, bad_tile_size)

    w = None
    if weights is not None:
        w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
        if w.size != color_count:
            raise ValueError("weights must hold width*height values")

    opts = _native.QuantizationOptions(bool(dither), bool(palette_only), int(color_space), int(kmeans_niter),
                                       int(kmeans_max_samples), bool(verbose))
    # The reference transposes into planar R|G|B here (np.asfortranarray, patolette.pyx:388-391): ~60 ms for 16 MP.
    # An F-ordered float64 array goes to the planar entry as is; anything else is handed over row-major (at most a
    # dtype cast) and the first kernel reads it with stride 3.
    if colors.dtype == np.float64 and colors.flags.f_contiguous and not colors.flags.c_contiguous:
        data, entry = colors, "patolette_amd_quantize"
    else:
        data, entry = np.ascontiguousarray(colors, dtype=np.float64), "patolette_amd_quantize_rows"
    palette = np.zeros((palette_size, 3), dtype=np.float64, order='F')
    palette_map = None
    if not opts.palette_only:
        palette_map = np.zeros(width * height, dtype=np.uintp)
    exit_code = C.c_int(0)
    L = _native.lib()
    if verbose and tile_size > 0 and w is None:
        print('patolette ======== Generating saliency map')     # patolette.pyx:408-409
    # tile_size > 0 and no explicit weights: saliency weights derived on the device (patolette.pyx:407-414)
    getattr(L, entry)(width, height, _dp(data), _dp(w), float(tile_size), palette_size, C.byref(opts), _dp(palette),
                             palette_map.ctypes.data_as(_native.zp) if palette_map is not None and palette_map.size > 0 else None,
                             C.byref(exit_code))
    success = exit_code.value == 0
    message = L.get_patolette_exit_code_info_message(exit_code.value).decode('UTF-8')
    _raise_saliency(exit_code.value, message)
    if not success:
        return (success, None, None, message)
    if opts.palette_only:
        return (success, palette, None, message)
    return (success, palette, palette_map, message)


def set_kmeans_update(mode):
    """KMeans centroid update: 0 (default) = the reference's sequential f32 chains, bit for bit; 1 = order-free exact sums rounded
    once: faster and content-independent, but NOT the reference's result -- ~1e-6 of the colour range per iteration, ~1e-4 after
    the default 32 iterations at ~1000 members per centroid (outside the 1e-5 parity target), ~0.02 % of the index map follows
    (tests/test_gpu_kmeans_update.py).  Process-wide; returns the previous setting (include/patolette_amd.h:
    patolette_amd_set_kmeans_update).  Ignored for palettes beyond 4096 entries (the exact update runs)."""
    return int(_native.lib().patolette_amd_set_kmeans_update(int(mode)))


def saliency_weights(width, height, colors, tile_size=512):
    """The weights `quantize` derives when tile_size > 0 (reference `get_weights`, patolette.pyx:203-313),
    computed on the GPU.  colors as for `quantize`.  Returns width*height float64."""
    colors = np.asarray(colors)
    if colors.ndim != 2 or colors.shape[1] != 3 or colors.shape[0] != width * height:
        raise ValueError(color_mismatch)
    data = np.asfortranarray(colors, dtype=np.float64)
    out = np.zeros(width * height, dtype=np.float64)
    L = _native.lib()
    rc = L.patolette_amd_saliency_weights(width, height, _dp(data), float(tile_size), _dp(out))
    if rc == -2:
        _raise_saliency(-5, L.get_patolette_exit_code_info_message(-5).decode('UTF-8'))
    if rc == -3:
        _raise_saliency(-6, L.get_patolette_exit_code_info_message(-6).decode('UTF-8'))
    if rc != 0:
        raise RuntimeError(_native.last_error())
    return out


def quantize_u8(image, palette_size, dither=True, palette_only=False, color_space=ColorSpace_ICtCp, tile_size=512,
                kmeans_niter=32, kmeans_max_samples=512 ** 2, weights=None, want_quantized=True):
    """8-bit adaptor (additive): `image` is an (H, W, 3|4) uint8 sRGB array as an image
    decoder returns it, or a torch CUDA tensor of that shape (then nothing but the palettes crosses PCIe).  Does on the GPU what callers of the reference do by hand around `quantize`
    (README.md:147-194): `colors = img/255`, `palette_u8 = clip(palette*255).astype(uint8)` and
    `quantized = palette_u8[palette_map]`; 3 bytes per pixel cross PCIe instead of 24.

    Returns (success, palette_u8 (K,3) uint8, palette_map (H,W) uint8|uint16|uint32 or None,
    quantized (H,W,3) uint8 or None, palette (K,3) float64 as `quantize` returns it, message)."""
    if hasattr(image, "data_ptr") and getattr(image, "is_cuda", False):
        return _quantize_u8_torch(image, palette_size, dither, palette_only, color_space, tile_size, kmeans_niter,
                                  kmeans_max_samples, weights, want_quantized)
    img = np.ascontiguousarray(image)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] not in (3, 4):
        raise ValueError("image must be an (H, W, 3|4) uint8 array")
    height, width, channels = img.shape
    n = width * height
    w = None
    if weights is not None:
        w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
        if w.size != n:
            raise ValueError("weights must hold width*height values")
    opts = _native.QuantizationOptions(bool(dither), bool(palette_only), int(color_space), int(kmeans_niter),
                                       int(kmeans_max_samples), False)
    palette = np.zeros((palette_size, 3), dtype=np.float64, order='F')
    palette_u8 = np.zeros((max(palette_size, 0), 3), dtype=np.uint8)
    map_dtype = np.uint8 if palette_size <= 256 else (np.uint16 if palette_size <= 65536 else np.uint32)
    pmap = None if palette_only else np.zeros((height, width), dtype=map_dtype)
    quant = np.zeros((height, width, 3), dtype=np.uint8) if (want_quantized and not palette_only) else None
    code = C.c_int(0)
    L = _native.lib()
    vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None and a.size > 0 else None   # noqa: E731
    L.patolette_amd_u8(width, height, vp(img), channels, _dp(w), float(tile_size), palette_size, C.byref(opts), _dp(palette),
                       vp(palette_u8), vp(pmap), np.dtype(map_dtype).itemsize, vp(quant), C.byref(code))
    message = L.get_patolette_exit_code_info_message(code.value).decode('UTF-8')
    _raise_saliency(code.value, message)
    if code.value != 0:
        return (False, None, None, None, None, message)
    return (True, palette_u8, pmap, quant, palette, message)


def _quantize_u8_torch(image, palette_size, dither, palette_only, color_space, tile_size, kmeans_niter, kmeans_max_samples,
                       weights, want_quantized):
    """`quantize_u8` for a torch CUDA uint8 tensor (H, W, 3|4): the image, the index map (uint8 for K <= 256, else int32)
    and the reconstructed image stay in HBM (`patolette_amd_u8_device`); the palettes come back as numpy arrays.
    torch is only used to allocate the outputs and to order this call after the producer of `image`.
    Import torch BEFORE patolette_amd in such a process: both link a HIP runtime with the same SONAME and torch does not
    initialise on the one this library would otherwise load first."""
    import torch
    if image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] not in (3, 4):
        raise ValueError("image must be an (H, W, 3|4) uint8 tensor")
    img = image.contiguous()
    height, width, channels = (int(v) for v in img.shape)
    n = width * height
    dev = img.device
    w = None
    if weights is not None:
        w = torch.as_tensor(weights, dtype=torch.float64, device=dev).reshape(-1).contiguous()
        if w.numel() != n:
            raise ValueError("weights must hold width*height values")
    opts = _native.QuantizationOptions(bool(dither), bool(palette_only), int(color_space), int(kmeans_niter),
                                       int(kmeans_max_samples), False)
    palette = np.zeros((palette_size, 3), dtype=np.float64, order='F')
    palette_u8 = np.zeros((max(palette_size, 0), 3), dtype=np.uint8)
    me = 1 if palette_size <= 256 else 4
    pmap = None if palette_only else torch.zeros((height, width), dtype=torch.uint8 if me == 1 else torch.int32, device=dev)
    quant = torch.zeros((height, width, 3), dtype=torch.uint8, device=dev) if (want_quantized and not palette_only) else None
    code = C.c_int(0)
    L = _native.lib()
    with torch.cuda.device(dev):
        torch.cuda.current_stream().synchronize()          # the library runs on its own stream
        L.patolette_amd_u8_device(width, height, C.c_void_p(img.data_ptr()), channels,
                                  C.c_void_p(w.data_ptr()) if w is not None else None, float(tile_size), palette_size,
                                  C.byref(opts), _dp(palette), palette_u8.ctypes.data_as(C.c_void_p),
                                  C.c_void_p(pmap.data_ptr()) if pmap is not None else None, me,
                                  C.c_void_p(quant.data_ptr()) if quant is not None else None, C.byref(code))
    message = L.get_patolette_exit_code_info_message(code.value).decode('UTF-8')
    _raise_saliency(code.value, message)
    if code.value != 0:
        return (False, None, None, None, None, message)
    return (True, palette_u8, pmap, quant, palette, message)


def quantize_batch(width, height, images, palette_size, weights=None, dither=True, palette_only=False,
                   color_space=ColorSpace_ICtCp, tile_size=512, kmeans_niter=32, kmeans_max_samples=512 ** 2, verbose=False):
    """Quantise a list of independent images of identical size on the current GPU through
    `patolette_amd_batch` (up to six images in flight: uploads and host-side work of one image
    overlap kernels of another).  Per-image results are identical to separate `quantize` calls
    with the same arguments (batch extension).  `weights`: None or one entry (array or None) per image;
    images without explicit weights get the saliency-derived ones when tile_size > 0, as in `quantize`.
    Returns a list of `quantize` tuples."""
    if tile_size < 0:
        return [(False, None, None, bad_tile_size)] * len(images)
    count = len(images)
    n = width * height
    # all planar (F-ordered float64) -> the planar entry as is; otherwise every image goes row-major (no transposes)
    arrs = [np.asarray(im) for im in images]
    planar = all(a.ndim == 2 and a.dtype == np.float64 and a.flags.f_contiguous and not a.flags.c_contiguous for a in arrs)
    datas = arrs if planar else [np.ascontiguousarray(a, dtype=np.float64) for a in arrs]
    for d in datas:
        if d.ndim != 2 or d.shape[1] != 3:
            raise ValueError(bad_channel_count.format(d.shape[1] if d.ndim == 2 else "?"))
        if d.shape[0] != n:
            raise ValueError(color_mismatch)
    if weights is not None and len(weights) != count:
        raise ValueError("weights must hold one entry (array or None) per image")
    ws = [None] * count if weights is None else [None if w is None else np.ascontiguousarray(w, dtype=np.float64).reshape(-1)
                                                  for w in weights]
    for w in ws:
        if w is not None and w.size != n:
            raise ValueError("weights must hold width*height values")
    opts = _native.QuantizationOptions(bool(dither), bool(palette_only), int(color_space), int(kmeans_niter),
                                       int(kmeans_max_samples), bool(verbose))
    pals = [np.zeros((palette_size, 3), dtype=np.float64, order='F') for _ in range(count)]
    maps = [None if palette_only else np.zeros(n, dtype=np.uintp) for _ in range(count)]
    PD = _native.dp * count
    PZ = _native.zp * count
    d_arr = PD(*[_dp(d) for d in datas])
    w_arr = None if weights is None else PD(*[_dp(w) if w is not None else _native.dp() for w in ws])
    p_arr = PD(*[_dp(p) for p in pals])
    m_arr = None if palette_only else PZ(*[m.ctypes.data_as(_native.zp) for m in maps])
    codes = (C.c_int * count)()
    L = _native.lib()
    (L.patolette_amd_batch if planar else L.patolette_amd_batch_rows)(count, width, height, d_arr, w_arr, float(tile_size), palette_size,
                                                                      C.byref(opts), p_arr, m_arr, codes)
    out = []
    for i in range(count):
        msg = L.get_patolette_exit_code_info_message(codes[i]).decode('UTF-8')
        _raise_saliency(codes[i], msg)
        if codes[i] != 0:
            out.append((False, None, None, msg))
        else:
            out.append((True, pals[i], maps[i], msg))
    return out


def quantize_u8_batch(images, palette_size, weights=None, dither=True, palette_only=False, color_space=ColorSpace_ICtCp,
                      tile_size=512, kmeans_niter=32, kmeans_max_samples=512 ** 2, want_quantized=True):
    """`quantize_u8` for a list of (H, W, 3|4) uint8 images of identical shape through `patolette_amd_batch_u8`: up to six
    images in flight on the current GPU, and 3 bytes per pixel over PCIe instead of 24, so a host-fed batch is bound by the
    kernels rather than by the upload.  Per-image results are identical to separate `quantize_u8` calls.
    Returns a list of `quantize_u8` tuples."""
    if tile_size < 0:
        return [(False, None, None, None, None, bad_tile_size)] * len(images)
    imgs = [np.ascontiguousarray(im) for im in images]
    count = len(imgs)
    if count == 0:
        return []
    shape = imgs[0].shape
    for im in imgs:
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] not in (3, 4) or im.shape != shape:
            raise ValueError("images must be (H, W, 3|4) uint8 arrays of one shape")
    height, width, channels = shape
    n = width * height
    if weights is not None and len(weights) != count:
        raise ValueError("weights must hold one entry (array or None) per image")
    ws = [None] * count if weights is None else [None if w is None else np.ascontiguousarray(w, dtype=np.float64).reshape(-1) for w in weights]
    for w in ws:
        if w is not None and w.size != n:
            raise ValueError("weights must hold width*height values")
    opts = _native.QuantizationOptions(bool(dither), bool(palette_only), int(color_space), int(kmeans_niter),
                                       int(kmeans_max_samples), False)
    map_dtype = np.uint8 if palette_size <= 256 else (np.uint16 if palette_size <= 65536 else np.uint32)
    pals = [np.zeros((palette_size, 3), dtype=np.float64, order='F') for _ in range(count)]
    pal8 = [np.zeros((max(palette_size, 0), 3), dtype=np.uint8) for _ in range(count)]
    maps = [None if palette_only else np.zeros((height, width), dtype=map_dtype) for _ in range(count)]
    quants = [np.zeros((height, width, 3), dtype=np.uint8) if (want_quantized and not palette_only) else None for _ in range(count)]
    vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None and a.size > 0 else C.c_void_p()   # noqa: E731
    PV = C.c_void_p * count
    PD = _native.dp * count
    codes = (C.c_int * count)()
    L = _native.lib()
    L.patolette_amd_batch_u8(count, width, height, PV(*[vp(im) for im in imgs]), channels,
                             None if weights is None else PD(*[_dp(w) if w is not None else _native.dp() for w in ws]),
                             float(tile_size), palette_size, C.byref(opts), PD(*[_dp(p) for p in pals]), PV(*[vp(p) for p in pal8]),
                             None if palette_only else PV(*[vp(m) for m in maps]), np.dtype(map_dtype).itemsize,
                             PV(*[vp(q) for q in quants]) if (want_quantized and not palette_only) else None, codes)
    out = []
    for i in range(count):
        msg = L.get_patolette_exit_code_info_message(codes[i]).decode('UTF-8')
        _raise_saliency(codes[i], msg)
        if codes[i] != 0:
            out.append((False, None, None, None, None, msg))
        else:
            out.append((True, pal8[i], maps[i], quants[i], pals[i], msg))
    return out


__all__ = [
    "__doc__",
    "__version__",
    "quantize",
    "quantize_batch",
    "quantize_u8",
    "quantize_u8_batch",
    "saliency_weights",
    "ColorSpace_sRGB",
    "ColorSpace_CIELuv",
    "ColorSpace_ICtCp",
]
