#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the MI355X-native patolette path (BASELINE.json metric).

A "step" = one full quantisation (convert -> GQ -> LQ -> KMeans -> palette map) of one synthetic
image whose f64 planar pixels are ALREADY RESIDENT IN HBM when the timed region starts
(patolette_amd_device; the PCIe-inclusive host-to-host rate is in DESIGN.md, never `value`).
Default workload = the configuration the metric is quoted on ("256-color ICtCp + KMeans"):
BASELINE.json configs[2], 4096x4096, K=256, ICtCp, KMeans 32 it / 512^2 samples, dither off.

  python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c3full] [--no-cpu-baseline]

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); images are independent,
so ranks shard the batch with NO data-path collective ("scaling": "weak": every rank quantises
its own image per step); barrier + max-over-ranks timing; rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (width, height, K, color_space, kmeans_niter, kmeans_max_samples, dither, weighted, description)
    "c2": (1920, 1080, 256, 2, 0, 512 ** 2, False, False, "BASELINE configs[1]: 1920x1080, 256 colors, ICtCp, KMeans off, dither off"),
    "c3": (4096, 4096, 256, 2, 32, 512 ** 2, False, False, "BASELINE configs[2]: 4096x4096, 256 colors, ICtCp + KMeans (32 it, 512^2 samples), dither off"),
    "c3full": (4096, 4096, 256, 2, 32, 4096 * 4096, False, False, "configs[2] stress: as c3 but kmeans_max_samples = N (all pixels clustered)"),
    "c4": (8192, 8192, 256, 1, 0, 512 ** 2, True, True, "BASELINE configs[3]: 8192x8192, 256 colors, CIELuv + weights + Riemersma dither"),
    "c4map": (8192, 8192, 256, 1, 0, 512 ** 2, False, True, "configs[3] without dither: 8192x8192, 256 colors, CIELuv + weights, NN map"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--force-dist", action="store_true", help="take the torch.distributed (RCCL) path even with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    dist = None
    torch = None
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)

    from patolette_amd import _native
    L = _native.lib()
    if L.patolette_amd_device_count() <= 0:
        raise SystemExit("bench.py: no HIP device visible (the HIP path has no CPU fallback)")
    if L.patolette_amd_set_device(local_rank) != 0:
        raise SystemExit("bench.py: cannot select device %d" % local_rank)

    width, height, K, cs, niter, max_samples, dither, weighted, desc = CONFIGS[args.config]
    n = width * height
    opts = _native.QuantizationOptions(dither, False, cs, niter, max_samples, False)
    pool = max(1, min(3, args.steps))
    d_imgs, d_wts = [], []
    for i in range(pool):
        p = L.patolette_amd_malloc(3 * n * 8)
        if not p:
            raise SystemExit("bench.py: hipMalloc failed")
        assert L.patolette_amd_fill_image(p, n, 100 * rank + i) == 0
        d_imgs.append(p)
        if weighted:
            q = L.patolette_amd_malloc(n * 8)
            assert L.patolette_amd_fill_weights(q, n, 100 * rank + i) == 0
            d_wts.append(q)
    import numpy as np
    pals = np.zeros((args.steps + args.warmup, K, 3), dtype=np.float64)
    pal = np.zeros((K, 3), dtype=np.float64, order="F")
    code = C.c_int(0)
    if dist is not None:
        # index maps (u8, K <= 256) of every step stay in HBM in a torch tensor the library writes into
        # directly, so the final RCCL gather needs no staging copy
        maps_t = torch.empty((args.steps, n), dtype=torch.uint8, device="cuda")
        warm_t = torch.empty((n,), dtype=torch.uint8, device="cuda")
        map_ptr = lambda i: (maps_t[i - args.warmup].data_ptr() if i >= args.warmup else warm_t.data_ptr())
    else:
        d_map = L.patolette_amd_malloc(n)      # K <= 256 -> u8 index map left in HBM
        map_ptr = lambda i: d_map

    def step(i):
        L.patolette_amd_device(width, height, d_imgs[i % pool], d_wts[i % pool] if weighted else None, K, C.byref(opts),
                               pal.ctypes.data_as(_native.dp), map_ptr(i), 1, C.byref(code))
        if code.value != 0:
            raise SystemExit("bench.py: quantisation failed: %s" % _native.last_error())
        pals[i] = pal

    def barrier():
        L.patolette_amd_synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    if not args.no_profile:
        _native.profile(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    L.patolette_amd_synchronize()
    gathered = None
    if dist is not None:
        # the only collective of the job: final gather of the results to rank 0 over RCCL/xGMI
        pal_t = torch.from_numpy(pals[args.warmup:]).to("cuda")
        gl_m = [torch.empty_like(maps_t) for _ in range(world)] if rank == 0 else None
        gl_p = [torch.empty_like(pal_t) for _ in range(world)] if rank == 0 else None
        dist.gather(maps_t, gl_m, dst=0)
        dist.gather(pal_t, gl_p, dst=0)
        torch.cuda.synchronize()
        gathered = (gl_m, gl_p)
    elapsed = time.perf_counter() - t0
    stats = _native.last_stats()
    prof = _native.profile_results() if not args.no_profile else {}
    _native.profile(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_px = float(n) * args.steps * world
    value = total_px / elapsed / 1e6
    # ---- roofline of the dominant kernel (most accumulated time in the timed region) ----
    roofline = None
    kernels = {}
    if prof:
        for name, r in prof.items():
            if r["launches"] and r["total_ms"] > 0:
                kernels[name] = {"ms_per_step": r["total_ms"] / args.steps, "launches_per_step": r["launches"] / args.steps,
                                 "GBps": r["bytes"] / (r["total_ms"] * 1e-3) / 1e9}
        dom = max(prof, key=lambda k: prof[k]["total_ms"])
        r = prof[dom]
        achieved = r["bytes"] / (r["total_ms"] * 1e-3) / 1e9
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "avg_launch_us": round(1e3 * r["total_ms"] / r["launches"], 3),
                    "algorithmic_bytes_per_launch": round(r["bytes"] / r["launches"], 1),
                    "time_share": round(r["total_ms"] / max(1e-9, sum(v["total_ms"] for v in prof.values())), 3)}

    # ---- CPU baseline: the oracle (plain-C port of the reference algorithm), one core, bounded sample ----
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        from oracle import binding as ob
        sw, sh = (2048, 1536) if n > 2048 * 1536 else (width, height)
        if dither and n > 1024 * 1024:
            sw, sh = 1024, 1024
        sn = sw * sh
        flat = ob.image(sn, 0)
        wt = ob.weights(sn, 0) if weighted else None
        t1 = time.perf_counter()
        ec, _, _ = ob.patolette(sw, sh, flat, wt, K, dither=dither, color_space=cs, kmeans_niter=niter,
                                kmeans_max_samples=min(max_samples, sn) if max_samples > 512 ** 2 else max_samples)
        dt = time.perf_counter() - t1
        cpu = {"value": round(sn / dt / 1e6, 4), "unit": "Mpx/s", "cores": 1, "kind": "port",
               "sample": "oracle (plain-C restatement of the reference path, single thread) on %dx%d of the same workload, %.1f s; stages %s"
                         % (sw, sh, dt, {k: round(v, 2) for k, v in ob.last_timings().items()})}

    out = {
        "metric": "Mpixels/sec quantized (256-color ICtCp + KMeans) at 1 GPU" if args.config.startswith("c3") else "Mpixels/sec quantized",
        "value": round(value, 3), "unit": "Mpx/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": desc, "width": width, "height": height, "palette_size": K, "images_per_step_per_gpu": 1,
                   "input": "uniform random sRGB (splitmix64), planar f64, resident in HBM; index map left in HBM as u8",
                   "kernel_events_in_timed_region": not args.no_profile,
                   "final_gather": ("RCCL gather of u8 maps + f64 palettes to rank 0, inside the timed region" if dist is not None else "none (1 GPU)")},
        "roofline": roofline, "cpu_baseline": cpu,
        "stages_ms_last_step": {k: round(v, 3) for k, v in stats.items() if k.startswith("ms_")},
        "run": {k: v for k, v in stats.items() if not k.startswith("ms_")},
        "kernels": {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_step"])},
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
