"""Randomised end-to-end parity sweep (GPU box): small images of varied shape / content / options through
patolette_amd.quantize and through the oracle; reports every case whose palette or map differs."""
import sys
import numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
from oracle import binding as ob
from tests.util import scene, match_rows_up_to_permutation

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
detail = set(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else set()
bad = 0
for case in range(ncases):
    h, w = int(rng.integers(1, 90)), int(rng.integers(1, 90))
    n = h * w
    kind = rng.choice(["noise", "scene", "few", "flat", "gradient", "u8"])
    if kind == "noise":
        colors = rng.random((n, 3))
    elif kind == "scene" and h > 4 and w > 4:
        colors = scene(h, w, int(rng.integers(0, 1000))).reshape(-1, 3)
    elif kind == "few":
        pal = rng.random((int(rng.integers(1, 6)), 3))
        colors = pal[rng.integers(0, len(pal), size=n)]
    elif kind == "flat":
        colors = np.tile(rng.random(3), (n, 1))
    elif kind == "gradient":
        t = np.linspace(0, 1, n)[:, None]
        colors = np.clip(t * rng.random(3) + (1 - t) * rng.random(3), 0, 1)
    else:
        colors = rng.integers(0, 256, size=(n, 3)).astype(np.float64) / 255
    colors = np.ascontiguousarray(colors)
    K = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 256, 300]))
    cs = int(rng.integers(0, 3))
    dither = bool(rng.integers(0, 2))
    niter = int(rng.choice([0, 0, 1, 3]))
    weighted = bool(rng.integers(0, 2))
    wts = (1.0 + rng.random(n) * rng.choice([0.0, 3.0, 1000.0])) if weighted else None
    ok, pal_g, map_g, msg = p.quantize(w, h, colors, K, dither=dither, color_space=cs, tile_size=0, kmeans_niter=niter,
                                       kmeans_max_samples=int(rng.choice([1, 4096, 512 ** 2])), weights=wts)
    ec, pal_o, map_o = ob.patolette(w, h, ob.planar(colors), wts, K, dither=dither, color_space=cs, kmeans_niter=niter,
                                    kmeans_max_samples=512 ** 2)
    desc = "case %d %dx%d %s K=%d cs=%d dither=%d niter=%d weighted=%d" % (case, w, h, kind, K, cs, dither, niter, weighted)
    if (ec == 0) != ok:
        print("STATUS MISMATCH", desc, ec, ok, msg); bad += 1; continue
    if not ok:
        continue
    same_pal = np.allclose(pal_g, pal_o, rtol=0, atol=1e-9, equal_nan=True)
    same_map = np.array_equal(map_g, map_o)
    if not (same_pal and same_map):
        # equivalent results?  the reconstructed images must agree even when the palette order differs (degenerate
        # clusters: eigenvector signs of a rank-deficient covariance are rounding noise in the reference too)
        rec_g, rec_o = pal_g[map_g], pal_o[map_o]
        rdiff = float(np.max(np.abs(rec_g - rec_o)))
        rows_g = sorted(map(tuple, np.round(pal_g[pal_g[:, 0] >= 0], 9).tolist()))
        rows_o = sorted(map(tuple, np.round(pal_o[pal_o[:, 0] >= 0], 9).tolist()))
        same_set = rows_g == rows_o
        tag = "REORDERED" if (same_set and rdiff <= 1e-9) else ("SAME-SET, maps differ" if same_set else "DIFF")
        print(tag, desc, "reconstruction maxdiff %.3g" % rdiff, "palette sets equal:", same_set,
              "distinct colours %d" % len(np.unique(np.round(colors, 12), axis=0)))
        if tag == "DIFF":
            bad += 1
        if case in detail:
            np.set_printoptions(precision=6, suppress=True, linewidth=200)
            print("  stats gpu:", {k: v for k, v in p.last_stats().items() if not k.startswith("ms_")})
            used_g, used_o = pal_g[pal_g[:, 0] >= -0.5], pal_o[pal_o[:, 0] >= -0.5]
            print("  gpu palette rows %d, oracle rows %d" % (len(used_g), len(used_o)))
            print("  gpu sorted:\n", np.array(sorted(map(tuple, used_g.tolist())))[:12])
            print("  oracle sorted:\n", np.array(sorted(map(tuple, used_o.tolist())))[:12])
            r = ob.quantize_clusters(ob.convert(["srgb_to_rec2020", "srgb_to_cieluv", "srgb_to_ictcp"][cs], ob.planar(colors)) if cs else ob.planar(colors), wts, n, K)
            print("  oracle clusters:", r["n_clusters"], "base", r["n_base"], "evals", r["split_evals"])
print("cases %d, bad %d" % (ncases, bad))
