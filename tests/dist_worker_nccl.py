"""Worker of tests/test_gpu_dist.py: the batch-sharding entry on the RCCL backend (`nccl`), one process per GPU.
Rank r binds to GPU LOCAL_RANK, quantises its block through patolette_amd_batch_dmap (maps left in HBM) and the
results are gathered to rank 0 from device memory.  Rank 0 compares every image with a separate `quantize` call."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (before patolette_amd: both link a HIP runtime)
import torch.distributed as dist  # noqa: E402


def main():
    out_path, side, count, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl")
    rank = dist.get_rank()
    import ctypes as C
    import patolette_amd as p
    from patolette_amd import _native
    from patolette_amd import dist as pdist
    L = _native.lib()
    assert L.patolette_amd_set_device(local_rank) == 0
    w = h = side
    n = w * h
    # the batch, identical on every rank: synthetic images generated in HBM and brought to the host
    d = L.patolette_amd_malloc(3 * n * 8)
    images = []
    for i in range(count):
        assert L.patolette_amd_fill_image(d, n, 300 + i) == 0
        flat = np.empty(3 * n)
        assert L.patolette_amd_memcpy_d2h(flat.ctypes.data_as(C.c_void_p), d, flat.nbytes) == 0
        images.append(flat.reshape(3, n).T)                       # (n,3) F-ordered view: the planar layout, no copy
    L.patolette_amd_free(d)
    kw = dict(dither=False, tile_size=0, kmeans_niter=32)
    res = pdist.quantize_batch_sharded(w, h, images, K, dist=dist, narrow_maps=True, **kw)
    # default tile_size (saliency weights) and 8-bit images on two small pictures
    rng = np.random.default_rng(5)
    small = [rng.integers(0, 256, size=(96, 128, 3), dtype=np.uint8) for _ in range(3)]
    res8 = pdist.quantize_batch_sharded(128, 96, small, 24, dist=dist, dither=True, kmeans_niter=2)
    # a failing image must not hang the collective: a flat image has a singular border covariance (exit code -6)
    bad = [small[0], np.full((96, 128, 3), 7, dtype=np.uint8)]
    resb = pdist.quantize_batch_sharded(128, 96, bad, 24, dist=dist, dither=False, kmeans_niter=0)
    # ... nor may a malformed one (two channels; wrong size) or a loader that raises: they fail before the library is called
    def loader(i):
        if i == 2:
            raise OSError("cannot decode image 2")
        return [small[0], small[1][:, :, :2], None, small[2][:50]][i]
    resm = pdist.quantize_batch_sharded(128, 96, loader, 24, dist=dist, count=4, dither=False, kmeans_niter=0)
    ok = True
    notes = []
    if rank == 0:
        ok = len(res) == count and all(r[0] for r in res)
        for i, r in enumerate(res):
            one = p.quantize(w, h, images[i], K, **kw)
            same = one[0] and np.array_equal(r[1], one[1]) and r[2].dtype == np.uint8 and np.array_equal(r[2], one[2])
            notes.append("image %d %s" % (i, "same" if same else "DIFFERS"))
            ok = ok and same
        for im, r in zip(small, res8):
            one = p.quantize_u8(im, 24, dither=True, kmeans_niter=2)
            same = r[0] and np.array_equal(r[1], one[4]) and np.array_equal(r[2], one[2].reshape(-1))
            notes.append("u8 %s" % ("same" if same else "DIFFERS"))
            ok = ok and same
        okb = resb[0][0] and (not resb[1][0]) and resb[1][1] is None and "singular" in resb[1][3].lower()
        notes.append("failure tuple %s: %s" % ("ok" if okb else "WRONG", resb[1][3]))
        ok = ok and okb
        one = p.quantize_u8(small[0], 24, dither=False, kmeans_niter=0)
        okm = [r[0] for r in resm] == [True, False, False, False] and "cannot decode image 2" in resm[2][3] and \
            np.array_equal(resm[0][1], one[4]) and np.array_equal(resm[0][2], one[2].reshape(-1))
        notes.append("malformed images %s: %s" % ("ok" if okm else "WRONG", [r[3] for r in resm]))
        ok = ok and okm
        with open(out_path, "w") as f:
            f.write(("OK\n" if ok else "MISMATCH\n") + "\n".join(notes))
    else:
        assert res is None and res8 is None and resb is None and resm is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
