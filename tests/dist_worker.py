"""Worker of tests/test_dist.py: world_size-2 gloo run of the batch-sharding path on CPU.
The per-image quantiser is the CPU oracle (test infrastructure): what is under test is the
sharding, ordering and gather logic of patolette_amd/dist.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from oracle import binding as ob  # noqa: E402
from patolette_amd import dist as pdist  # noqa: E402


def oracle_quantize(width, height, colors, palette_size, weights=None, dither=True, palette_only=False, color_space=2,
                    tile_size=0, kmeans_niter=32, kmeans_max_samples=512 ** 2, verbose=False):
    flat = ob.planar(colors)
    ec, pal, pmap = ob.patolette(width, height, flat, weights, palette_size, dither=dither, palette_only=palette_only,
                                 color_space=color_space, kmeans_niter=kmeans_niter, kmeans_max_samples=kmeans_max_samples)
    return (ec == 0, pal if ec == 0 else None, pmap if ec == 0 else None, "Quantization successful." if ec == 0 else "error")


def main():
    out_path = sys.argv[1]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w, h, K, count = 40, 30, 12, 5                       # 5 images over 2 ranks: uneven shards (3 + 2)
    n = w * h
    images = [ob.image(n, 40 + i).reshape(3, n).T.copy() for i in range(count)]
    res = pdist.quantize_batch_sharded(w, h, images, K, dist=dist, quantize_fn=oracle_quantize, dither=False, kmeans_niter=0)
    # a failing image must not strand the other rank in the gathers: image 1 (rank 0's block) is malformed, the loader of
    # image 4 (rank 1's block) raises; both come back as failures with the reason, the rest as usual
    def loader(i):
        if i == 4:
            raise OSError("cannot decode image 4")
        return images[i][:, :2] if i == 1 else images[i]

    def checked_quantize(width, height, colors, palette_size, **kw):
        if np.asarray(colors).shape != (width * height, 3):
            raise ValueError("bad shape %s" % (np.asarray(colors).shape,))
        return oracle_quantize(width, height, colors, palette_size, **kw)
    res_bad = pdist.quantize_batch_sharded(w, h, loader, K, dist=dist, quantize_fn=checked_quantize, count=count, dither=False,
                                           kmeans_niter=0)
    # timing protocol of bench.py: barrier, then max-over-ranks of a per-rank scalar
    import torch
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == float(world)
    # within-image sharding (tests/split_model.py): each rank holds half of one node's pixels; the all-reduced moment table and
    # the cut must equal, bit for bit, what the whole node gives in one process
    from tests import split_model as psplit
    rng = np.random.default_rng(77)
    npx = 30001
    c = rng.random((npx, 3))
    wts = 1.0 + 2.0 * rng.random(npx)
    axis = np.array([0.6, -0.3, 0.74])
    bink = psplit.make_bink(2, 15)
    lo, hi = pdist.shard(npx, rank, world)
    hi += lo
    reduce_fn = lambda a, op: psplit.allreduce_dist(dist, a, op)      # noqa: E731
    cut, _, red = psplit.split_node_sharded(c[lo:hi], wts[lo:hi], axis, bink, reduce_fn)
    cut1, _, red1 = psplit.split_node_sharded(c, wts, axis, bink, lambda a, op: a)
    split_ok = cut == cut1 and np.array_equal(red["parts"].view(np.uint64), red1["parts"].view(np.uint64)) and \
        np.array_equal(red["count"], red1["count"]) and np.array_equal(red["size"], red1["size"])
    # the collective patolette_amd_slice borrows (patolette_amd.dist.make_comm), called through its C function pointer on
    # host staging as the library does with a gloo group: SUM of f64 / i64 / i32, and the one-row-per-rank table that
    # carries minima, maxima and exclusive prefixes
    import ctypes as C
    comm = pdist.make_comm(dist)
    comm_ok = comm.rank == rank and comm.size == world and comm.host_buffers == 1
    for dtype, npdt in ((0, np.float64), (1, np.int64), (2, np.int32)):
        a = (np.arange(7) + 10 * rank).astype(npdt)
        rc = comm.allreduce_sum(None, a.ctypes.data_as(C.c_void_p), a.size, dtype)
        comm_ok = comm_ok and rc == 0 and np.array_equal(a, sum((np.arange(7) + 10 * r) for r in range(world)).astype(npdt))
    table = np.zeros((world, 3), dtype=np.uint64)
    table[rank] = [np.uint64(0xFFFFFFFFFFFFFF00 + rank), np.uint64(5 + rank), np.uint64(100 * (rank + 1))]
    rc = comm.allreduce_sum(None, table.ctypes.data_as(C.c_void_p), table.size, 1)
    comm_ok = comm_ok and rc == 0 and int(table[:, 0].min()) == 0xFFFFFFFFFFFFFF00 and int(table[:, 1].max()) == 5 + world - 1 and \
        int(table[:rank, 2].sum()) == sum(100 * (r + 1) for r in range(rank))
    split_ok = split_ok and comm_ok
    flag = torch.tensor([1.0 if split_ok else 0.0], dtype=torch.float64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    split_ok = float(flag.item()) == 1.0
    if rank == 0:
        assert len(res) == count
        single = [oracle_quantize(w, h, im, K, dither=False, kmeans_niter=0) for im in images]
        ok = all(r[0] and np.array_equal(r[1], s[1]) and np.array_equal(r[2], s[2]) for r, s in zip(res, single))
        ok = ok and [r[0] for r in res_bad] == [True, False, True, True, False]
        ok = ok and "bad shape" in res_bad[1][3] and "cannot decode image 4" in res_bad[4][3]
        ok = ok and all(np.array_equal(res_bad[i][1], single[i][1]) and np.array_equal(res_bad[i][2], single[i][2]) for i in (0, 2, 3))
        shards = [pdist.shard(count, r, world) for r in range(world)]
        with open(out_path, "w") as f:
            f.write("OK" if ok and split_ok and shards == [(0, 3), (3, 2)] else "MISMATCH")
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
