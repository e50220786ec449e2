"""Batch sharding of independent images over the GPUs of one node.

Images are independent (the reference itself is one-image-per-call), so every rank runs the
full single-GPU pipeline on a contiguous block of the batch and there is NO collective on the
data path.  torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests) is
used only for the final gather of results to rank 0: palettes (K*3 f64 = 6 KB/image) and index
maps as u8 when K <= 256 (1 B/px; never size_t over the fabric).  A direct gather to one root
uses the 7 distinct inbound xGMI links of that GPU, which a ring all-gather would not.
"""
import numpy as np


def shard(count, rank, world):
    """Contiguous block [start, start+n) of `count` items owned by `rank`."""
    base, rem = divmod(count, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def gather_to_root(local, dist, root=0, device=None):
    """Gather equally-shaped per-rank numpy arrays (first axis = local batch, padded to the
    largest shard) to `root`; returns the list of per-rank arrays on root, None elsewhere."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    t = torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device))
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    if t.shape[0] < mx:
        pad = torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], dim=0)
    out = [torch.empty_like(t) for _ in range(world)] if rank == root else None
    dist.gather(t, out, dst=root)
    if rank != root:
        return None
    return [o[:n].cpu().numpy() for o, n in zip(out, sizes)]


def quantize_batch_sharded(width, height, images, palette_size, dist=None, quantize_fn=None, weights=None, **kwargs):
    """Quantise `images` (a list, identical on every rank, or a callable i -> (N,3) float64 array or (H,W,3|4) uint8 image) with
    the batch sharded over the ranks of `dist`; rank 0 returns [(success, palette, map, message)]
    for the whole batch in order, the other ranks return None.  With dist=None it is a plain
    loop.  `quantize_fn` defaults to patolette_amd.quantize (tests inject the CPU oracle to
    exercise the sharding and gather logic without a GPU)."""
    batch_fn = None
    if quantize_fn is None:
        from . import quantize as quantize_fn
        from . import quantize_batch as batch_fn          # three images in flight per GPU
    count = len(images) if not callable(images) else kwargs.pop("count")
    get = images if callable(images) else (lambda i: images[i])
    kwargs.setdefault("tile_size", 0)
    rank, world = (0, 1) if dist is None else (dist.get_rank(), dist.get_world_size())
    start, n = shard(count, rank, world)
    local = []
    if batch_fn is not None:
        for g0 in range(start, start + n, 6):                 # groups of six bound the host memory held at once
            idx = list(range(g0, min(g0 + 6, start + n)))
            ws = None if weights is None else [weights[i] for i in idx]
            group = [get(i) for i in idx]
            if all(getattr(im, "dtype", None) == np.uint8 and getattr(im, "ndim", 0) == 3 for im in group):
                # 8-bit images as decoded, (H, W, 3|4): 3 bytes per pixel over PCIe instead of 24
                from . import quantize_u8_batch
                kw = {k: v for k, v in kwargs.items() if k != "verbose"}
                for r in quantize_u8_batch(group, palette_size, weights=ws, want_quantized=False, **kw):
                    local.append((r[0], r[4], None if r[2] is None else r[2].reshape(-1), r[5]))
            else:
                local.extend(batch_fn(width, height, group, palette_size, weights=ws, **kwargs))
    else:
        for i in range(start, start + n):
            w = None if weights is None else weights[i]
            local.append(quantize_fn(width, height, get(i), palette_size, weights=w, **kwargs))
    if dist is None:
        return local
    npx = width * height
    ok = np.array([1 if r[0] else 0 for r in local], dtype=np.int64).reshape(-1, 1)
    pals = np.stack([np.asarray(r[1], dtype=np.float64) if r[1] is not None else np.full((palette_size, 3), np.nan) for r in local]) \
        if local else np.zeros((0, palette_size, 3))
    mdt = np.uint8 if palette_size <= 256 else np.int32
    have_map = not kwargs.get("palette_only", False)
    maps = np.stack([np.asarray(r[2]).astype(mdt) if r[2] is not None else np.zeros(npx, dtype=mdt) for r in local]) \
        if local else np.zeros((0, npx), dtype=mdt)
    g_ok = gather_to_root(ok, dist)
    g_pal = gather_to_root(pals, dist)
    g_map = gather_to_root(maps, dist) if have_map else None
    if rank != 0:
        return None
    out = []
    for r in range(world):
        for j in range(g_ok[r].shape[0]):
            success = bool(g_ok[r][j, 0])
            msg = "Quantization successful." if success else "Internal quantization error."
            if not success:
                out.append((False, None, None, msg))
            else:
                out.append((True, np.asfortranarray(g_pal[r][j]), g_map[r][j].astype(np.uintp) if have_map else None, msg))
    return out
