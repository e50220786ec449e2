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


# ---- K = 256: the configuration every BASELINE config uses (254 greedy commits, deep candidate trees) ----------------
@pytest.mark.parametrize("kind,n,weighted,seed", [("ictcp", 65536, False, 31), ("luv", 70000, True, 32), ("blobs", 65536, True, 33),
                                                  ("modes", 66000, False, 34)])
def test_gq_lq_at_256_colours_matches_independent_reading(ob, kind, n, weighted, seed):
    """As above at K = 256 on >= 65 536 colours: every one of the ~254 greedy commits (palette ORDER), the membership of every
    colour and the centres."""
    K = 256
    c = dataset(kind, n, 300 + seed, ob)
    w = (1.0 + 3.0 * np.random.default_rng(seed).random(n) ** 3) if weighted else None
    centers, member, nbase = ir.quantize_clusters(c, w, K)
    got = ob.quantize_clusters(np.ascontiguousarray(c.T).reshape(-1), w, n, K)
    assert got["rc"] == 0 and got["n_base"] == nbase and got["n_clusters"] == len(centers) == K
    assert np.array_equal(got["member"].astype(np.int64), member)
    assert np.allclose(got["centers"][:K], centers, rtol=0, atol=1e-12 * max(1.0, np.abs(centers).max()))


# ---- the orchestrator: stage sequencing and colour-space routing of lib/src/patolette.c:157-343 ----------------------
def _pinned_stages(ob):
    """The two stages reference builds pin in this image (colour code: libref_color.so; KMeans: libref_faiss.so), through the
    oracle's entry points that tests/test_oracle_pinning.py holds to those builds bit for bit."""
    def conv(name, a):
        n = a.shape[0]
        return ob.convert(name, np.ascontiguousarray(a.T).reshape(-1)).reshape(3, n).T.copy()

    def kmeans(c, w, centers, niter, max_samples):
        return ob.kmeans_refine(np.ascontiguousarray(c.T).reshape(-1), w, c.shape[0], np.ascontiguousarray(centers), niter, max_samples)
    return conv, kmeans


@pytest.mark.parametrize("niter", [0, 3])
@pytest.mark.parametrize("palette_only", [False, True])
@pytest.mark.parametrize("dither", [False, True])
@pytest.mark.parametrize("color_space", [0, 1, 2])
def test_orchestrator_routing_matches_independent_reading(ob, color_space, dither, palette_only, niter):
    """All three colour spaces x dither / NN map x palette_only x KMeans on/off: the CIELuv -> Rec2020 -> sRGB -> ICtCp detour of
    the NN map, the ICtCp back-conversion the sRGB palette goes through all the same (patolette.c:322-323), palette_only
    returning the palette in the quantisation space, the -1 rows."""
    w_, h_, K = 37, 29, 24
    n = w_ * h_
    seed = 500 + 7 * color_space + 3 * int(dither) + int(palette_only) + 11 * niter
    flat = ob.image(n, seed)
    colors = flat.reshape(3, n).T.copy()
    weights = (1.0 + 2.0 * np.random.default_rng(seed).random(n)) if (seed % 2) else None
    conv, kmeans = _pinned_stages(ob)
    want = ir.patolette(w_, h_, colors, weights, K, dither, palette_only, color_space, niter, 512 ** 2, conv, kmeans)
    assert want is not None
    ec, pal, pmap = ob.patolette(w_, h_, flat, weights, K, dither=dither, palette_only=palette_only, color_space=color_space,
                                 kmeans_niter=niter, kmeans_max_samples=512 ** 2)
    assert ec == 0
    assert np.allclose(pal, want[0], rtol=0, atol=1e-12 * max(1.0, np.abs(want[0]).max()))
    if palette_only:
        assert pmap is None and want[1] is None
    else:
        assert np.array_equal(pmap.astype(np.int64), want[1])


def test_orchestrator_more_colours_than_pixels_and_unset_rows(ob):
    """K > pixel count: the clusters run out, the unset palette rows are -1 (patolette.c:327-336)."""
    w_, h_, K = 5, 4, 40
    n = w_ * h_
    flat = ob.image(n, 77)
    conv, kmeans = _pinned_stages(ob)
    want = ir.patolette(w_, h_, flat.reshape(3, n).T.copy(), None, K, False, False, 2, 0, 512 ** 2, conv, kmeans)
    ec, pal, pmap = ob.patolette(w_, h_, flat, None, K, dither=False, color_space=2, kmeans_niter=0)
    assert ec == 0 and np.array_equal(pal == -1.0, want[0] == -1.0) and (pal == -1.0).any()
    # the last splits are of two-colour clusters: rank-1 covariance, the eigenvector SIGN -- which child is "left" -- is rounding
    # noise that differs between LAPACK builds, so two palette rows may trade places: same rows, same image
    from tests.util import match_rows_up_to_permutation
    assert match_rows_up_to_permutation(pal, want[0], 1e-12) is not None
    assert np.array_equal(pal[pmap], want[0][want[1]])
