import sys, time, numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
rng = np.random.default_rng(0)
imgs = [np.asfortranarray(rng.random((4096 * 4096, 3))) for _ in range(6)]
for it in range(3):
    out = None
    t = time.time(); out = p.quantize_batch(4096, 4096, imgs, 256, dither=False, tile_size=0); a = time.time() - t
    print("batch of 6, host to host: %.1f ms / image, %.0f Mpx/s" % (a * 1e3 / 6, 6 * 16.777216 / a), all(o[0] for o in out))
imgs8 = [rng.integers(0, 256, size=(4096, 4096, 3), dtype=np.uint8) for _ in range(6)]
for it in range(3):
    out = None
    t = time.time(); out = p.quantize_u8_batch(imgs8, 256, dither=False, tile_size=0); a = time.time() - t
    print("u8 batch of 6, host to host (u8 map + reconstructed image back): %.1f ms / image, %.0f Mpx/s" % (a * 1e3 / 6, 6 * 16.777216 / a), all(o[0] for o in out))
for it in range(2):
    out = None
    t = time.time(); out = p.quantize_u8_batch(imgs8, 256, dither=False, tile_size=0, want_quantized=False); a = time.time() - t
    print("u8 batch of 6, host to host (u8 map back): %.1f ms / image, %.0f Mpx/s" % (a * 1e3 / 6, 6 * 16.777216 / a), all(o[0] for o in out))
