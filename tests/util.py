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
