"""ONE image dealt out over the GPUs of a node (patolette_amd_slice_device): strong scaling of a single quantisation.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29590 \
        tools/slice_bench.py [--side 4096] [--colors 256] [--steps 8] [--warmup 2]

Every rank holds 1/N of the pixels in HBM; the per-node reductions cross the group as RCCL all-reduces on the library's
device buffers (patolette_amd.dist.make_comm).  Rank 0 prints one JSON line: whole-image Mpx/s (max over ranks of the
time), the per-stage times of its last call and how many collectives a call made.  With N = 1 the line shows what the
protocol itself costs on one GPU (the same kernels, five stream synchronisations per split round instead of one).
Not part of bench.py's contract: the driver's line is the batch-sharded workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--colors", type=int, default=256)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kmeans-niter", type=int, default=32)
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29590")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    import numpy as np
    from patolette_amd import _native
    from patolette_amd import dist as pdist
    L = _native.lib()
    assert L.patolette_amd_set_device(local_rank) == 0
    n = args.side * args.side
    K = args.colors
    whole = torch.empty(3 * n, dtype=torch.float64, device="cuda")
    assert L.patolette_amd_fill_image(C.c_void_p(whole.data_ptr()), n, 0) == 0       # the same image on every rank
    torch.cuda.synchronize()
    b, c = pdist.shard(n, rank, world)
    mine = whole.view(3, n)[:, b:b + c].contiguous()                                # planar x | y | z of the slice
    del whole
    dmap = torch.empty(c, dtype=torch.uint8 if K <= 256 else torch.int32, device="cuda")
    calls = [0]
    comm = pdist.make_comm(dist)
    inner = comm._keepalive

    def counted(ctx, buf, count, dtype):
        calls[0] += 1
        return inner(ctx, buf, count, dtype)
    cb = _native.ALLREDUCE_SUM_FN(counted)
    comm.allreduce_sum = cb
    opts = _native.QuantizationOptions(False, False, 2, args.kmeans_niter, 512 ** 2, False)
    pal = np.zeros((K, 3), order="F")
    code = C.c_int(0)

    def one():
        L.patolette_amd_slice_device(n, b, c, C.c_void_p(mine.data_ptr()), None, K, C.byref(opts), C.byref(comm),
                                     pal.ctypes.data_as(_native.dp), C.c_void_p(dmap.data_ptr()), 1 if K <= 256 else 4, C.byref(code))
        assert code.value == 0, code.value
    for _ in range(args.warmup):
        one()
    dist.barrier()
    torch.cuda.synchronize()
    calls[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    stats = _native.last_stats()
    if rank == 0:
        print(json.dumps({"metric": "Mpixels/sec, ONE %dx%d image over the group (256-color ICtCp + KMeans)" % (args.side, args.side),
                          "value": round(n * args.steps / elapsed / 1e6, 3), "unit": "Mpx/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "scaling": "strong", "dtype": "f64",
                          "data": "synthetic", "collectives_per_image": calls[0] // args.steps,
                          "stages_ms_last_step": {k: round(v, 3) for k, v in stats.items() if k.startswith("ms_")},
                          "run": {k: v for k, v in stats.items() if not k.startswith("ms_")}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
