import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def match_rows_up_to_permutation(got, ref, atol):
    """Bijection perm with got[perm[i]] ~= ref[i] (NaN rows match NaN rows); None if there is none.

    Used only for inputs whose clusters are exactly collinear (two distinct colours): their
    covariance is rank-1, the eigenvector SIGN dsyev returns is then decided by rounding noise
    (it differs between LAPACK builds too, SURVEY.md 7 hard part 1b), so left/right -- hence
    palette ORDER -- is not defined by the reference; the palette as a set still is."""
    used = set()
    perm = []
    for r in ref:
        hit = None
        for j, g in enumerate(got):
            if j in used:
                continue
            if np.allclose(g, r, rtol=0, atol=atol, equal_nan=True):
                hit = j
                break
        if hit is None:
            return None
        used.add(hit)
        perm.append(hit)
    return np.array(perm)


def scene(rows, cols, seed):
    """Synthetic 'photograph' (rows, cols, 3) float64 in [0,1]: smooth gradients, a few saturated blobs and mild
    noise -- the kind of input the saliency stage is meant for (noise images have no salient region)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
    img = np.stack([0.5 + 0.4 * np.sin(xx / (0.15 * cols + 3.0) + seed),
                    0.5 + 0.4 * np.cos(yy / (0.12 * rows + 3.0)),
                    0.35 + 0.25 * np.sin((xx + yy) / (0.2 * (rows + cols) + 3.0))], axis=2)
    for _ in range(3):
        cy, cx = rng.uniform(0.25, 0.75) * rows, rng.uniform(0.25, 0.75) * cols
        ry, rx = rng.uniform(0.05, 0.18) * rows + 1, rng.uniform(0.05, 0.18) * cols + 1
        inside = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
        img[inside] = rng.uniform(0.0, 1.0, size=3)
    return np.clip(img + rng.normal(0.0, 0.02, img.shape), 0.0, 1.0)
