"""Per-kernel times of one 4096 x 4096 quantisation (K = 256, ICtCp, KMeans) for different image CONTENT: uniform noise (what
bench.py uses), a smooth synthetic scene, a posterised one (few distinct colours).  LDS histograms and scatter ranks see
very different collision patterns."""
import sys
import numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
from patolette_amd import _native

n = 4096
rng = np.random.default_rng(3)
y, x = np.mgrid[0:n, 0:n].astype(np.float32)
smooth = np.stack([0.5 + 0.5 * np.sin(x / 337.0) * np.cos(y / 253.0), (x + y) / (2.0 * n), 0.5 + 0.5 * np.cos((x - y) / 571.0)], axis=-1).reshape(-1, 3).astype(np.float64)
imgs = {"noise": rng.random((n * n, 3)), "smooth": smooth, "smooth+noise": np.clip(smooth + 0.02 * rng.standard_normal((n * n, 3)), 0, 1),
        "posterised": np.round(smooth * 7) / 7, "smooth, 8-bit": np.round(smooth * 255) / 255,
        "smooth+noise, 8-bit": np.round(np.clip(smooth + 0.02 * rng.standard_normal((n * n, 3)), 0, 1) * 255) / 255}
try:
    from tests.util import scene
    imgs["scene (tests.util)"] = scene(n, n, 4).reshape(-1, 3)
    imgs["scene, 8-bit"] = np.round(imgs["scene (tests.util)"] * 255) / 255
except Exception as ex:                                      # noqa: BLE001
    print("no scene generator:", ex)
only = sys.argv[1].split(";") if len(sys.argv) > 1 else None
for name, colors in imgs.items():
    if only and name not in only:
        continue
    p.quantize(n, n, colors, 256, dither=False, tile_size=0)
    p.profile(True)
    ok = p.quantize(n, n, colors, 256, dither=False, tile_size=0)[0]
    prof = p.profile_results()
    p.profile(False)
    st = _native.last_stats()
    top = sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:7]
    print("%-20s device %.2f ms (convert %.2f gq %.2f lq %.2f km %.2f map %.2f), rounds %d evals %d | " % (
        name, st["ms_total"] - st["ms_upload"] - st["ms_download"], st["ms_convert"], st["ms_gq"], st["ms_lq"], st["ms_kmeans"], st["ms_map"],
        st["lq_rounds"], st["split_evals"]) + ", ".join("%s %.2f" % (k, v["total_ms"]) for k, v in top))
