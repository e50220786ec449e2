"""The oracle against an INDEPENDENT second reading of the reference (tests/independent_ref.py: numpy written from the C
sources, LAPACK dsyev / BLAS dgemv from scipy's OpenBLAS): the stages no reference build can pin in this image (GQ, LQ,
NN map, Riemersma walk + dither) -- DESIGN.md section 2.  CPU only."""
import numpy as np
import pytest

from tests import independent_ref as ir


def dataset(kind, n, seed, ob):
    rng = np.random.default_rng(seed)
    if kind == "ictcp":                                     # what the pipeline feeds the quantiser: converted uniform noise
        return ob.convert("srgb_to_ictcp", ob.image(n, seed)).reshape(3, n).T.copy()
    if kind == "luv":
        return ob.convert("srgb_to_cieluv", ob.image(n, seed)).reshape(3, n).T.copy()
    if kind == "blobs":                                     # a few anisotropic clusters: GQ keeps several base clusters
        k = int(rng.integers(3, 7))
        cen = rng.random((k, 3))
        a = rng.integers(0, k, size=n)
        return cen[a] + rng.standard_normal((n, 3)) * rng.uniform(0.005, 0.05, size=(k, 3))[a]
    if kind == "line":                                      # almost one-dimensional
        t = rng.random((n, 1))
        return t * np.array([[0.9, 0.4, 0.2]]) + rng.standard_normal((n, 3)) * 1e-3
    if kind == "modes":                                     # several modes strung along one direction, thin across: 3-8 base clusters
        k = int(rng.integers(3, 8))
        pos = np.sort(rng.random(k))[rng.integers(0, k, size=n)]
        d = np.array([[0.7, 0.5, 0.3]])
        return pos[:, None] * d + rng.standard_normal((n, 1)) * 0.01 * d + rng.standard_normal((n, 3)) * 2e-3
    raise ValueError(kind)


CASES = [(kind, n, K, weighted, seed)
         for seed, (kind, n, K, weighted) in enumerate([
             ("ictcp", 3000, 2, False), ("ictcp", 3000, 3, False), ("ictcp", 5000, 8, False), ("ictcp", 4000, 16, True),
             ("luv", 3000, 5, False), ("luv", 4000, 12, True), ("luv", 2500, 16, False),
             ("blobs", 3000, 4, False), ("blobs", 4000, 8, False), ("blobs", 5000, 13, True), ("blobs", 2000, 16, True),
             ("blobs", 6000, 24, False), ("blobs", 3500, 6, True), ("blobs", 3000, 12, False),
             ("line", 2000, 4, False), ("line", 3000, 9, True), ("line", 2500, 16, False),
             ("ictcp", 6000, 32, False), ("luv", 5000, 20, True), ("ictcp", 2000, 7, True),
             ("blobs", 4500, 10, False), ("luv", 3000, 3, True),
             ("modes", 3000, 8, False), ("modes", 4000, 16, True), ("modes", 2500, 12, False), ("modes", 3500, 5, True)])]


@pytest.mark.parametrize("kind,n,K,weighted,seed", CASES)
def test_gq_lq_matches_independent_reading(ob, kind, n, K, weighted, seed):
    """Global + local quantiser: same base-cluster count, same membership of every colour, same palette order, centres to
    1e-12 -- against numpy code that calls the real dsyev / dgemv and takes the first maximum of the split objective over
    all 512 cuts."""
    c = dataset(kind, n, 100 + seed, ob)
    w = (1.0 + 3.0 * np.random.default_rng(seed).random(n) ** 3) if weighted else None
    want = ir.quantize_clusters(c, w, K)
    assert want is not None
    centers, member, nbase = want
    got = ob.quantize_clusters(np.ascontiguousarray(c.T).reshape(-1), w, n, K)
    assert got["rc"] == 0 and got["n_base"] == nbase and got["n_clusters"] == len(centers)
    assert np.array_equal(got["member"].astype(np.int64), member)
    assert np.allclose(got["centers"][:len(centers)], centers, rtol=0, atol=1e-12 * max(1.0, np.abs(centers).max()))


@pytest.mark.parametrize("kind,seed,weighted", [("ictcp", 1, False), ("luv", 2, True), ("blobs", 3, False), ("blobs", 4, True)])
def test_split_decision_is_not_a_rounding_artefact(ob, kind, seed, weighted):
    """The cut the f64 objective picks is also the arg-max of the objective evaluated in long double over all 512 cuts."""
    n = 4000
    c = dataset(kind, n, 200 + seed, ob)
    w = (1.0 + 2.0 * np.random.default_rng(seed).random(n)) if weighted else None
    idx = np.arange(n)
    a = ir.split_cluster(c, w, idx)
    b = ir.split_cluster(c, w, idx, extended=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_nn_map_matches_numpy_brute_force(ob):
    for n, k, seed in [(20000, 256, 1), (5000, 7, 2), (3000, 64, 3)]:
        flat = ob.convert("srgb_to_ictcp", ob.image(n, seed))
        pal = ob.convert("srgb_to_ictcp", ob.image(k, 50 + seed)).reshape(3, k).T.copy()
        pal[k // 2] = pal[k // 3]                           # exact tie: the lower index wins
        assert np.array_equal(ob.nn_map(flat, n, pal).astype(np.int64), ir.nn_map(flat.reshape(3, n).T, pal))


@pytest.mark.parametrize("w,h", [(1, 1), (2, 2), (5, 3), (8, 8), (13, 7), (16, 9), (33, 40), (64, 64), (100, 3)])
def test_hilbert_walk_matches_direct_transcription(ob, w, h):
    want = ir.hilbert_walk(w, h)
    got = ob.hilbert_order(w, h)
    assert [int(v) for v in got] == [y * w + x for (x, y) in want]
    if max(w, h) > 1:
        assert sorted(int(v) for v in got) == list(range(w * h))       # every pixel exactly once


@pytest.mark.parametrize("w,h,k,seed", [(17, 11, 8, 1), (32, 32, 16, 2), (40, 25, 64, 3), (9, 50, 5, 4)])
def test_dither_matches_independent_reading(ob, w, h, k, seed):
    n = w * h
    flat = ob.image(n, seed)
    img = flat.reshape(3, n).T.copy()
    pal = ob.image(k, 70 + seed).reshape(3, k).T.copy()
    want = ir.dither(img, w, h, pal)
    got = ob.dither(flat, w, h, pal).astype(np.int64)
    assert np.array_equal(got, want)
