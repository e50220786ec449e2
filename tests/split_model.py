"""Within-image sharding over several GPUs: the exchange protocol in numpy.

The HIP path implements this protocol as `patolette_amd_slice` (pipeline.hip; Python: `patolette_amd.dist.quantize_image_sharded`;
`-m gpu` tests: tests/test_gpu_slice.py).  This module is its host-side model: it needs no GPU, so tests/test_dist.py can pin down
-- on two gloo ranks and on arbitrary dealings of the pixels -- the property that makes the sharding EXACT:

Every per-node reduction of the split loop (projection extrema, per-bucket moments, children's centred moments) is either an
integer, an ordered-key minimum / maximum, or a sum of two-part "binned" addends that lie on fixed grids (devutil.h
`bin_split`: with |v| <= 2^E and <= 2^P addends the parts are multiples of 2^(E-B) and 2^(E-2B), B = 51 - P, so partial sums
never leave the 53-bit significand).  Such sums are exact in ANY order and grouping, so an all-reduce (SUM / MIN / MAX) of the
ranks' partial tables gives bit for bit the table one GPU computes over the whole image -- no matter how the pixels are dealt
out -- and every rank then takes the same cut (`k_cut`'s arithmetic on the reduced table) without further communication.
Per split round that is one all-reduce of 2 ordered keys per node (16 B) and one of <= (4 x 2 x 512 f64 + 512 u64 + 512 u32) per
node (39 KB); the pixels never move, each rank partitions its own slice.

The functions mirror the device code they stand for (file:line in patolette_amd/csrc); numpy, host memory.
"""
import numpy as np

BUCKETS = 512          # common.h kBuckets
DELTA = 1e-16          # common.h kDelta


def make_bink(E, P):
    """devutil.h make_bink: the two magic constants of the grids for |v| <= 2^E, <= 2^P addends."""
    B = min(40, max(8, 51 - P))
    return np.ldexp(1.5, 52 + E - B), np.ldexp(1.5, 52 + E - 2 * B)


def bin_split(v, bink):
    """devutil.h bin_split: v -> (v0 on the coarse grid, v1 = the remainder on the fine grid)."""
    M0, M1 = bink
    v0 = (v + M0) - M0
    r = v - v0
    v1 = (r + M1) - M1
    return v0, v1


def f64_key(d):
    """devutil.h f64_key: monotone double -> uint64, so extrema reduce with integer MIN / MAX."""
    u = np.asarray(d, dtype=np.float64).view(np.uint64)
    neg = (u >> np.uint64(63)).astype(bool)
    return np.where(neg, ~u, u | np.uint64(1 << 63))


def key_f64(k):
    k = np.asarray(k, dtype=np.uint64)
    pos = (k >> np.uint64(63)).astype(bool)
    return np.where(pos, k & np.uint64((1 << 63) - 1), ~k).view(np.float64)


def local_extrema(colors, axis):
    """k_minmax (quant.hip): ordered keys of the projection extrema of this rank's pixels of the node."""
    if colors.shape[0] == 0:
        return np.array([np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64(0)], dtype=np.uint64)
    dots = (colors[:, 0] * axis[0] + colors[:, 1] * axis[1]) + colors[:, 2] * axis[2]
    return np.array([f64_key(dots.min()), f64_key(dots.max())], dtype=np.uint64)


def buckets_of(colors, axis, mn, mx, first_slot=0):
    """k_hist's bucket rule (sort.c:61-87): round-robin over the node's pixel slots when degenerate."""
    if mx - mn < DELTA:
        return ((first_slot + np.arange(colors.shape[0])) % BUCKETS).astype(np.int64)
    dots = (colors[:, 0] * axis[0] + colors[:, 1] * axis[1]) + colors[:, 2] * axis[2]
    ratio = (dots - mn) * (1 / (mx - mn))
    return np.minimum((BUCKETS * ratio).astype(np.int64), BUCKETS - 1)


def local_tables(colors, weights, axis, mn, mx, bink, first_slot=0):
    """k_hist<W, false> (quant.hip): per bucket the two binned parts of sum(w c) (3) and sum(w), the pixel count and
    the truncated weight sum (local.c:133).  Returns dict of arrays: parts (4, 2, 512) f64, count (512,) u64, size (512,) u64."""
    b = buckets_of(colors, axis, mn, mx, first_slot)
    w = np.ones(colors.shape[0]) if weights is None else weights
    parts = np.zeros((4, 2, BUCKETS))
    for q in range(4):
        v = w if q == 3 else colors[:, q] * w
        v0, v1 = bin_split(v, bink)
        parts[q, 0] = np.bincount(b, weights=v0, minlength=BUCKETS)      # exact in any order: the addends lie on a grid
        parts[q, 1] = np.bincount(b, weights=v1, minlength=BUCKETS)
    count = np.bincount(b, minlength=BUCKETS).astype(np.uint64)
    size = count if weights is None else np.bincount(b, weights=np.floor(w), minlength=BUCKETS).astype(np.uint64)
    return dict(parts=parts, count=count, size=size), b


def cut_of(tab, weighted):
    """k_cut (quant.hip): prefix sums, objective sum_j csl^2/sl + csr^2/sr (local.c:150-168), FIRST maximum."""
    p = np.cumsum(tab["parts"], axis=2)                                   # exact: parts lie on the grids
    siz = np.cumsum(tab["size"] if weighted else tab["count"]).astype(np.float64)
    obj = np.zeros(BUCKETS)
    sl, sr = siz, siz[-1] - siz
    for j in range(3):
        csl = p[j, 0] + p[j, 1]
        csr = (p[j, 0, -1] + p[j, 1, -1]) - csl
        v = np.zeros(BUCKETS)
        np.divide(csl * csl, sl, out=v, where=sl != 0)
        t = np.zeros(BUCKETS)
        np.divide(csr * csr, sr, out=t, where=sr != 0)
        obj += v + t
    return int(np.argmax(obj))


def allreduce_dist(dist, arr, op):
    """One all-reduce over a torch.distributed group (gloo on the CPU, nccl = RCCL on GPUs: the tables are tiny, 39 KB)."""
    import torch
    ops = {"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}
    if arr.dtype == np.uint64:                                            # torch has no uint64 reductions
        if op == "sum":                                                   # counts: far below 2^63, add as int64
            t = torch.from_numpy(arr.view(np.int64).copy())
            dist.all_reduce(t, op=ops[op])
            return t.numpy().view(np.uint64)
        t = torch.from_numpy((arr ^ np.uint64(1 << 63)).view(np.int64).copy())      # keys: an order-preserving map onto int64
        dist.all_reduce(t, op=ops[op])
        return t.numpy().view(np.uint64) ^ np.uint64(1 << 63)
    t = torch.from_numpy(np.ascontiguousarray(arr).copy())
    dist.all_reduce(t, op=ops[op])
    return t.numpy()


def split_node_sharded(colors, weights, axis, bink, reduce_fn, first_slot=0):
    """One split of one node whose pixels are dealt out over the ranks: `colors` / `weights` are THIS rank's pixels of the node
    (in image order), `reduce_fn(array, op)` the all-reduce.  Returns (cut bucket, this rank's bucket ids, reduced table)."""
    ext = local_extrema(colors, axis)
    mn = key_f64(reduce_fn(ext[:1], "min"))[0]
    mx = key_f64(reduce_fn(ext[1:], "max"))[0]
    tab, b = local_tables(colors, weights, axis, mn, mx, bink, first_slot)
    red = dict(parts=reduce_fn(tab["parts"], "sum"), count=reduce_fn(tab["count"], "sum"), size=reduce_fn(tab["size"], "sum"))
    return cut_of(red, weights is not None), b, red
