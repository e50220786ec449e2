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
    # timing protocol of bench.py: barrier, then max-over-ranks of a per-rank scalar
    import torch
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == float(world)
    if rank == 0:
        assert len(res) == count
        single = [oracle_quantize(w, h, im, K, dither=False, kmeans_niter=0) for im in images]
        ok = all(r[0] and np.array_equal(r[1], s[1]) and np.array_equal(r[2], s[2]) for r, s in zip(res, single))
        shards = [pdist.shard(count, r, world) for r in range(world)]
        with open(out_path, "w") as f:
            f.write("OK" if ok and shards == [(0, 3), (3, 2)] else "MISMATCH")
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
