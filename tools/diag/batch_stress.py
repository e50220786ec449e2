"""Stress of the batch entry on small images (device-driven split loop, several engines in flight): counts calls that fail or differ from the first result."""
import sys
import numpy as np
import patolette_amd as p

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(8)
imgs = [rng.integers(0, 256, size=(128, 128, 3), dtype=np.uint8) for _ in range(5)]
p.quantize_u8_batch(imgs, 32, dither=False, tile_size=64, kmeans_niter=2)
ref = None
bad = diff = 0
for r in range(reps):
    out = p.quantize_u8_batch(imgs[:2], 16, dither=False, palette_only=True, tile_size=0)
    if not all(o[0] for o in out):
        bad += 1
        continue
    pal = [o[1].copy() for o in out]
    if ref is None:
        ref = pal
    elif not all(np.array_equal(a, b) for a, b in zip(pal, ref)):
        diff += 1
print("reps %d failed %d differing %d" % (reps, bad, diff))
