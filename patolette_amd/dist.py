"""Batch sharding of independent images over the GPUs of one node.

Images are independent (the reference itself is one-image-per-call), so every rank runs the
full single-GPU pipeline on a contiguous block of the batch and there is NO collective on the
data path.  torch.distributed is used only for the final gather of results to rank 0: exit
codes, palettes (K*3 f64 = 6 KB/image) and index maps as u8 when K <= 256 (1 B/px; never
size_t over the fabric).  A direct gather to one root uses the 7 distinct inbound xGMI links
of that GPU, which a ring all-gather would not.

backend "nccl" (= RCCL over xGMI on ROCm): every rank binds its engine to GPU LOCAL_RANK, the
library leaves each image's index map in HBM (`patolette_amd_batch_dmap`) inside one torch
tensor per rank, and RCCL gathers exit codes, palettes and maps from device memory.
backend "gloo" (CPU tests, or GPUs without RCCL): the same protocol on host tensors.
Import torch (and initialise the process group) BEFORE the first patolette_amd call in such a
process: torch and libpatolette_amd.so link HIP runtimes of the same SONAME.
"""
import ctypes as C
import os

import numpy as np

MESSAGES = {0: "Quantization successful.", -1: "Internal quantization error."}


def shard(count, rank, world):
    """Contiguous block [start, start+n) of `count` items owned by `rank`."""
    base, rem = divmod(count, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def collective_device(dist):
    """torch.device the collectives of this process group need their tensors on (None = host)."""
    import torch
    if dist.get_backend() != "nccl":
        return None
    local_rank = int(os.environ.get("LOCAL_RANK", dist.get_rank() % max(1, torch.cuda.device_count())))
    return torch.device("cuda", local_rank)


def gather_to_root(local, dist, root=0, device=None):
    """Gather per-rank arrays or tensors (first axis = local batch, padded to the largest shard)
    to `root`; returns the list of per-rank numpy arrays on root, None elsewhere.  `device`:
    where the collective runs (collective_device(dist)); a tensor already there is gathered
    in place (the RCCL path hands in the device-resident maps)."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    t = local if isinstance(local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local))
    if device is not None and t.device != device:
        t = t.to(device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device))
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    if t.shape[0] < mx:
        pad = torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], dim=0)
    out = [torch.empty_like(t) for _ in range(world)] if rank == root else None
    dist.gather(t.contiguous(), out, dst=root)
    if rank != root:
        return None
    return [o[:n].cpu().numpy() for o, n in zip(out, sizes)]


def _check_weights(weights, count, npx):
    if weights is None:
        return None
    if len(weights) != count:
        raise ValueError("weights must hold one entry (array or None) per image")
    out = []
    for w in weights:
        if w is None:
            out.append(None)
            continue
        w = np.ascontiguousarray(w, dtype=np.float64).reshape(-1)
        if w.size != npx:
            raise ValueError("weights must hold width*height values")
        out.append(w)
    return out


def _quantize_block_to_device(width, height, get, idx, palette_size, weights, device, opts_kw, errors):
    """This rank's block through patolette_amd_batch_dmap: host images in, maps left in HBM in one torch tensor.
    Returns (codes int64 (n,1), palettes (n,K,3) f64, maps tensor (n, N) on `device`)."""
    import torch
    from . import _native, bad_channel_count, color_mismatch
    L = _native.lib()
    npx = width * height
    n = len(idx)
    me = 1 if palette_size <= 256 else 4
    palette_only = bool(opts_kw.get("palette_only", False))
    maps_t = torch.zeros((n, 0 if palette_only else npx), dtype=torch.uint8 if me == 1 else torch.int32, device=device)
    codes = np.full((n, 1), -1, dtype=np.int64)
    pals = np.full((n, palette_size, 3), np.nan)
    opts = _native.QuantizationOptions(bool(opts_kw.get("dither", True)), palette_only, int(opts_kw.get("color_space", 2)),
                                       int(opts_kw.get("kmeans_niter", 32)), int(opts_kw.get("kmeans_max_samples", 512 ** 2)), False)
    tile_size = float(opts_kw.get("tile_size", 512))
    torch.cuda.current_stream(device).synchronize()               # the library runs on its own streams
    def prepare(im):
        """(fmt, array) of one image as patolette_amd_batch_dmap takes it; ValueError with the binding's message if malformed."""
        im = np.asarray(im)
        if im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] in (3, 4):
            if im.shape[:2] != (height, width):
                raise ValueError("images must be (height, width, 3|4) uint8 arrays of one shape")
            return int(im.shape[2]), np.ascontiguousarray(im)
        planar = im.ndim == 2 and im.dtype == np.float64 and im.flags.f_contiguous and not im.flags.c_contiguous
        d = im if planar else np.ascontiguousarray(im, dtype=np.float64)
        if d.ndim != 2 or d.shape[1] != 3:
            raise ValueError(bad_channel_count.format(d.shape[1] if d.ndim == 2 else "?"))
        if d.shape[0] != npx:
            raise ValueError(color_mismatch)
        return (0 if planar else 1), d

    def run(js, fmt, datas):
        cnt = len(js)
        ws = None if weights is None else [weights[idx[j]] for j in js]
        gp = [np.zeros((palette_size, 3), dtype=np.float64, order="F") for _ in js]
        gc = (C.c_int * cnt)()
        PV, PD = C.c_void_p * cnt, _native.dp * cnt
        L.patolette_amd_batch_dmap(cnt, width, height, PV(*[d.ctypes.data_as(C.c_void_p) for d in datas]), fmt,
                                   None if ws is None else PD(*[w.ctypes.data_as(_native.dp) if w is not None else _native.dp() for w in ws]),
                                   tile_size, palette_size, C.byref(opts), PD(*[p.ctypes.data_as(_native.dp) for p in gp]),
                                   None if palette_only else C.c_void_p(maps_t[js[0]].data_ptr()), me, gc)
        for j, p, c in zip(js, gp, gc):
            codes[j, 0] = c
            if c == 0:
                pals[j] = p

    for g0 in range(0, n, 6):                                     # groups of six bound the host memory held at once
        # A malformed image (or a failing loader) must not raise here: the other ranks are on their way to the gathers and
        # would wait for this one for ever.  It gets exit code -1 (its message goes to `errors`), the rest of the block runs.
        ready = []
        for j in range(g0, min(g0 + 6, n)):
            try:
                ready.append((j,) + prepare(get(idx[j])))
            except Exception as e:                                # noqa: BLE001 -- whatever the loader or the checks raise
                errors[idx[j]] = "%s: %s" % (type(e).__name__, e)
        # maximal runs of consecutive images of one format go through one call (their maps are consecutive rows of maps_t)
        k = 0
        while k < len(ready):
            m = k + 1
            while m < len(ready) and ready[m][0] == ready[m - 1][0] + 1 and ready[m][1] == ready[k][1]:
                m += 1
            try:
                run([r[0] for r in ready[k:m]], ready[k][1], [r[2] for r in ready[k:m]])
            except Exception as e:                                # noqa: BLE001
                for r in ready[k:m]:
                    errors[idx[r[0]]] = "%s: %s" % (type(e).__name__, e)
            k = m
    return codes, pals, maps_t


def quantize_batch_sharded(width, height, images, palette_size, dist=None, quantize_fn=None, weights=None, narrow_maps=False,
                           **kwargs):
    """Quantise `images` (a list, identical on every rank, or a callable i -> (N,3) float64 array or (H,W,3|4) uint8 image,
    then with count=) with the batch sharded over the ranks of `dist`; rank 0 returns [(success, palette, map, message)]
    for the whole batch in order, the other ranks return None.  With dist=None it is a plain loop on the current GPU.
    Keyword arguments and their defaults are those of `patolette_amd.quantize` (tile_size=512 derives saliency weights
    for images without explicit `weights`, as there).  `narrow_maps` keeps the maps in their device width (uint8 for
    palette_size <= 256, else int32) instead of widening them to uintp on the root (256 maps of 16 MP are 4 GB narrow,
    34 GB wide).  A failing image does not raise: its tuple carries success=False and the exit-code message, so every
    rank reaches the collective.  `quantize_fn` replaces the per-image call (tests inject the CPU oracle to exercise
    the sharding and gather logic without a GPU)."""
    count = len(images) if not callable(images) else kwargs.pop("count")
    get = images if callable(images) else (lambda i: images[i])
    npx = width * height
    weights = _check_weights(weights, count, npx)
    rank, world = (0, 1) if dist is None else (dist.get_rank(), dist.get_world_size())
    start, n = shard(count, rank, world)
    idx = list(range(start, start + n))
    mdt = np.uint8 if palette_size <= 256 else np.int32
    have_map = not kwargs.get("palette_only", False)
    device = collective_device(dist) if dist is not None else None

    messages = dict(MESSAGES)
    if quantize_fn is None:
        from . import _native
        L = _native.lib()
        for c in range(-6, 1):
            m = L.get_patolette_exit_code_info_message(c)
            if m:
                messages[c] = m.decode("UTF-8")

    errors = {}                                                   # image number -> text of what it raised on THIS rank
    if quantize_fn is None and device is not None:
        # RCCL: bind this rank's engine to its GPU, leave the maps in HBM, gather from there
        if L.patolette_amd_set_device(device.index) != 0:
            raise RuntimeError("patolette_amd.dist: cannot bind to GPU %d" % device.index)
        codes, pals, maps = _quantize_block_to_device(width, height, get, idx, palette_size, weights, device, kwargs, errors)
    else:
        codes = np.full((n, 1), -1, dtype=np.int64)
        pals = np.full((n, palette_size, 3), np.nan)
        maps = np.zeros((n, npx if have_map else 0), dtype=mdt)

        def put(j, r, code=None):
            codes[j, 0] = (0 if r[0] else -1) if code is None else code
            if r[0]:
                pals[j] = np.asarray(r[1], dtype=np.float64)
                if have_map and r[2] is not None:
                    maps[j] = np.asarray(r[2]).reshape(-1).astype(mdt)
            elif code is None and len(r) > 3 and r[3] and r[3] != MESSAGES[-1]:
                errors[idx[j]] = r[3]

        def fail(j, e):
            # Nothing an image (or its loader) raises may leave this rank before the gathers: the others would wait for ever.
            # Only what the SALIENCY STAGE raises keeps its own exit code -- the binding turns codes -5 / -6 into a ValueError /
            # LinAlgError carrying the library's message for that code (patolette_amd._raise_saliency); a ValueError from a loader,
            # a shape check or an unknown keyword is the reference's "internal error" (-1) with its text, as on the RCCL path.
            text = str(e)
            code = -1
            if isinstance(e, np.linalg.LinAlgError) and messages.get(-6) and messages[-6] in text:
                code = -6
            elif isinstance(e, ValueError) and messages.get(-5) and text == messages[-5]:
                code = -5
            errors[idx[j]] = text if code != -1 else "%s: %s" % (type(e).__name__, e)
            put(j, (False, None, None, text), code)

        if quantize_fn is None:
            from . import quantize_batch, quantize_u8_batch

            def run_group(js, group):
                ws = None if weights is None else [weights[idx[j]] for j in js]
                if all(getattr(im, "dtype", None) == np.uint8 and getattr(im, "ndim", 0) == 3 for im in group):
                    # 8-bit images as decoded, (H, W, 3|4): 3 bytes per pixel over PCIe instead of 24
                    if any(im.shape[:2] != (height, width) for im in group):
                        raise ValueError("images must be (height, width, 3|4) uint8 arrays of one shape")
                    kw = {k: v for k, v in kwargs.items() if k != "verbose"}
                    return [(r[0], r[4], r[2], r[5]) for r in quantize_u8_batch(group, palette_size, weights=ws, want_quantized=False, **kw)]
                return quantize_batch(width, height, group, palette_size, weights=ws, **kwargs)

            for g0 in range(0, n, 6):                             # groups of six bound the host memory held at once
                grp, imgs = [], []
                for j in range(g0, min(g0 + 6, n)):               # a failing loader fails its own image, before any GPU work
                    try:
                        imgs.append(get(idx[j]))
                        grp.append(j)
                    except Exception as e:                        # noqa: BLE001
                        fail(j, e)
                if not grp:
                    continue
                try:
                    for j, r in zip(grp, run_group(grp, imgs)):
                        put(j, r)
                except Exception:                                 # noqa: BLE001 -- a malformed image: find it, one image at a time
                    for j, im in zip(grp, imgs):
                        try:
                            put(j, run_group([j], [im])[0])
                        except Exception as e:                    # noqa: BLE001
                            fail(j, e)
        else:
            for j, i in enumerate(idx):
                try:
                    w = None if weights is None else weights[i]
                    put(j, quantize_fn(width, height, get(i), palette_size, weights=w, **kwargs))
                except Exception as e:                            # noqa: BLE001
                    fail(j, e)

    # what a failing image raised travels with its exit code (fixed 240-byte rows), so the root reports the real reason
    texts = np.zeros((n, 240), dtype=np.uint8)
    for j, i in enumerate(idx):
        if i in errors:
            raw = errors[i].encode("UTF-8", "replace")[:240]
            texts[j, :len(raw)] = np.frombuffer(raw, dtype=np.uint8)

    if dist is None:
        g_code, g_pal, g_map, g_txt = [codes], [pals], [maps.cpu().numpy() if hasattr(maps, "cpu") else maps], [texts]
    else:
        g_code = gather_to_root(codes, dist, device=device)
        g_txt = gather_to_root(texts, dist, device=device)
        g_pal = gather_to_root(pals, dist, device=device)
        g_map = gather_to_root(maps, dist, device=device) if have_map else None
        if rank != 0:
            return None
    out = []
    for r in range(len(g_code)):
        for j in range(g_code[r].shape[0]):
            code = int(g_code[r][j, 0])
            msg = messages.get(code, MESSAGES[-1])
            if code != 0:
                why = bytes(g_txt[r][j]).rstrip(b"\0").decode("UTF-8", "replace")
                msg = why if why and code in (-5, -6) else (msg + " (" + why + ")" if why else msg)
                out.append((False, None, None, msg))
            else:
                m = None
                if have_map:
                    m = g_map[r][j] if narrow_maps else g_map[r][j].astype(np.uintp)
                out.append((True, np.asfortranarray(g_pal[r][j]), m, msg))
    return out


# --------------------------------------------------------------------------------------------
# ONE image over the GPUs of a group: every rank holds a contiguous slice of the pixels
# --------------------------------------------------------------------------------------------
_TORCH_DTYPES = {0: ("float64", "<f8", np.float64), 1: ("int64", "<i8", np.int64), 2: ("int32", "<i4", np.int32)}


class _DevMem:
    """A span of device memory the library lends to the collective, seen by torch through __cuda_array_interface__."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}


def make_comm(dist):
    """patolette_amd__Comm over a torch.distributed process group: the one collective `patolette_amd_slice` needs, an
    in-place element-wise SUM.  backend "nccl" (= RCCL): on the library's device buffers, no copies; any other backend
    (gloo): on the pinned host staging the library provides.  Keep the returned object alive during the call."""
    import torch
    from . import _native
    on_host = dist.get_backend() != "nccl"
    device = collective_device(dist)

    def allreduce(_ctx, ptr, count, dtype):
        try:
            name, typestr, npdt = _TORCH_DTYPES[dtype]
            if on_host:
                a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(npdt))), shape=(count,))
                t = torch.from_numpy(a)                      # shares the staging memory: reduced in place
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
            else:
                t = torch.as_tensor(_DevMem(ptr, count, typestr), device=device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize(device)               # the library reads the buffer from its own stream next
            return 0
        except Exception as ex:                              # never let an exception cross the C frame
            import sys
            print("patolette_amd.dist: all-reduce failed: %r" % (ex,), file=sys.stderr)
            return 1

    cb = _native.ALLREDUCE_SUM_FN(allreduce)
    comm = _native.Comm(dist.get_rank(), dist.get_world_size(), cb, None, 1 if on_host else 0)
    comm._keepalive = cb
    return comm


def quantize_image_sharded(total_pixels, slice_begin, slice_colors, palette_size, dist, weights=None, palette_only=False,
                           color_space=2, kmeans_niter=32, kmeans_max_samples=512 ** 2, verbose=False):
    """One image dealt out over the ranks of `dist`: this rank passes pixels
    [slice_begin, slice_begin + len(slice_colors)) of an image of `total_pixels` pixels -- `shard(total_pixels, rank, world)`
    gives the usual split -- as (n, 3) float64 sRGB, optionally its slice of explicit weights.  Returns
    (success, palette (K,3) F-ordered, slice_map (n,) uintp | None, message): the palette is the same on every rank and
    bit-identical to `patolette_amd.quantize(..., dither=False, tile_size=0)` of the whole image on one GPU after
    `patolette_amd_set_invariant_sums(1)`; slice_map is this rank's part of that image's index map (they stay where they are:
    gather them with `gather_to_root` if one rank needs the whole map).  Dithering and derived saliency weights are
    whole-image stages and not available per slice.  Import torch before patolette_amd in such a process."""
    from . import _native
    colors = np.asarray(slice_colors)
    if colors.ndim != 2 or colors.shape[1] != 3:
        raise ValueError("slice_colors must be (n, 3)")
    n = colors.shape[0]
    data = np.asfortranarray(colors, dtype=np.float64)       # planar x | y | z, as patolette() takes it
    w = None
    if weights is not None:
        w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
        if w.size != n:
            raise ValueError("weights must hold one value per pixel of the slice")
    opts = _native.QuantizationOptions(False, bool(palette_only), int(color_space), int(kmeans_niter), int(kmeans_max_samples),
                                       bool(verbose))
    comm = make_comm(dist)
    palette = np.zeros((palette_size, 3), dtype=np.float64, order="F")
    pmap = None if palette_only else np.zeros(n, dtype=np.uintp)
    code = C.c_int(0)
    L = _native.lib()
    L.patolette_amd_slice(int(total_pixels), int(slice_begin), n, data.ctypes.data_as(_native.dp) if n else None,
                          w.ctypes.data_as(_native.dp) if w is not None else None, int(palette_size), C.byref(opts), C.byref(comm),
                          palette.ctypes.data_as(_native.dp), pmap.ctypes.data_as(_native.zp) if pmap is not None and n else None,
                          C.byref(code))
    message = L.get_patolette_exit_code_info_message(code.value).decode("UTF-8")
    if code.value != 0:
        return (False, None, None, message)
    return (True, palette, pmap, message)
