"""Quantise a fixed set of images and dump palettes + maps + run statistics to an .npz (A/B of a library switch across two
processes: run twice with different environments, then `python tools/dump_results.py --compare a.npz b.npz`)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smooth(w, h):
    y, x = np.mgrid[0:h, 0:w]
    r = 0.5 + 0.5 * np.sin(x / 37.0) * np.cos(y / 53.0)
    g = (x + y) / float(w + h)
    b = 0.5 + 0.5 * np.cos((x - y) / 71.0)
    return np.stack([r, g, b], axis=-1).reshape(-1, 3)


def cases():
    rng = np.random.default_rng(11)
    yield "noise2048_K256", 2048, 2048, rng.random((2048 * 2048, 3)), None, 256, dict(color_space=2, kmeans_niter=0)
    yield "noise512_K256_km", 512, 512, rng.random((512 * 512, 3)), None, 256, dict(color_space=2, kmeans_niter=8)
    yield "smooth_K256", 1000, 800, smooth(1000, 800), None, 256, dict(color_space=2, kmeans_niter=0)
    yield "smooth_K37_luv_w", 640, 480, smooth(640, 480) * 0.8 + 0.2 * rng.random((640 * 480, 3)), 0.5 + 3 * rng.random(640 * 480), 37, dict(color_space=1, kmeans_niter=0)
    yield "srgb_K1000", 700, 700, rng.random((490000, 3)) ** 2.0, None, 1000, dict(color_space=0, kmeans_niter=0)
    blobs = np.concatenate([0.05 * rng.standard_normal((20000, 3)) + c for c in rng.random((12, 3))]).clip(0, 1)
    yield "blobs_K64", 240000, 1, blobs, None, 64, dict(color_space=2, kmeans_niter=0)
    yield "blobs_K7", 240000, 1, blobs, None, 7, dict(color_space=1, kmeans_niter=0)
    yield "tiny_K16", 9, 7, rng.random((63, 3)), None, 16, dict(color_space=2, kmeans_niter=0)


def main():
    if sys.argv[1] == "--compare":
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        ok = True
        for k in a.files:
            same = np.array_equal(a[k], b[k])
            if k.endswith("_stats"):
                print(k, a[k].tolist(), b[k].tolist())
                continue
            ok = ok and same
            if not same:
                print("DIFFERS", k)
        print("ALL EQUAL" if ok else "MISMATCH")
        return
    import patolette_amd as p
    from patolette_amd import _native
    out = {}
    for name, w, h, colors, weights, K, kw in cases():
        r = p.quantize(w, h, colors, K, dither=False, tile_size=0, weights=weights, **kw)
        assert r[0], name
        st = _native.last_stats()
        out[name + "_pal"] = r[1]
        out[name + "_map"] = r[2]
        out[name + "_stats"] = np.array([st["split_evals"], st["lq_rounds"], st["n_base_clusters"], st["n_clusters"]])
    np.savez(sys.argv[1], **out)


if __name__ == "__main__":
    main()
