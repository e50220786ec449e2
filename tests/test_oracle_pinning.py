"""The oracle against the golden vectors that pin it (CPU only).

color_ref.npz / kmeans_ref.npz were produced by builds of the reference's OWN sources
(oracle/_ref, see oracle/Makefile); eigen_lapack.npz by the LAPACK the reference calls.
When oracle/_ref is present (build container, GPU box) the live libraries are compared too.
"""
import ctypes as C
import os

import numpy as np
import pytest

from tests.util import ROOT, bits, golden  # noqa: F401
from tests.golden import make_golden as mg

CONV = ["srgb_to_ictcp", "srgb_to_cieluv", "srgb_to_rec2020", "ictcp_to_rec2020", "cieluv_to_rec2020", "rec2020_to_srgb"]
SRC = {"srgb_to_ictcp": None, "srgb_to_cieluv": None, "srgb_to_rec2020": None,
       "ictcp_to_rec2020": "srgb_to_ictcp", "cieluv_to_rec2020": "srgb_to_cieluv", "rec2020_to_srgb": "srgb_to_rec2020"}


@pytest.mark.parametrize("name", CONV)
def test_color_bit_exact_vs_reference_golden(ob, name):
    g = golden("color_ref.npz")
    src = mg.color_inputs(int(g["n"]), int(g["seed"])) if SRC[name] is None else g[SRC[name]]
    out = ob.convert(name, src)
    assert np.array_equal(bits(out), bits(g[name]))


def test_color_luv_to_ictcp_chain(ob):
    g = golden("color_ref.npz")
    x = ob.convert("srgb_to_ictcp", ob.convert("rec2020_to_srgb", ob.convert("cieluv_to_rec2020", g["srgb_to_cieluv"])))
    assert np.array_equal(bits(x), bits(g["cieluv_to_ictcp"]))


@pytest.mark.skipif(not os.path.exists(mg.REF_COLOR), reason="oracle/_ref not built here")
def test_color_bit_exact_vs_live_reference_build(ob):
    R = C.CDLL(mg.REF_COLOR)
    src = ob.image(50000, 77)
    for name in ["srgb_to_ictcp", "srgb_to_cieluv", "srgb_to_rec2020"]:
        assert np.array_equal(bits(ob.convert(name, src)), bits(mg.ref_convert(R, name, src)))


def test_eigen_vs_lapack_golden(ob):
    g = golden("eigen_lapack.npz")
    A, W, V = g["A"], g["W"], g["V"]
    bitexact = 0
    for a, w, v in zip(A, W, V):
        info, w2, v2 = ob.eigen_sym3(a)
        assert info == 0
        scale = max(1e-300, np.max(np.abs(w)))
        assert np.max(np.abs(w - w2)) <= 1e-13 * scale
        # principal axis (what the quantiser uses, pca.c:136-138) including SIGN
        assert np.max(np.abs(v[:, 2] - v2[:, 2])) < 1e-9, (a, v, v2)
        # full basis incl. signs whenever the spectrum is well separated
        gaps = np.diff(w)
        if np.all(gaps > 1e-6 * scale):
            assert np.max(np.abs(v - v2)) < 1e-8
        bitexact += np.array_equal(bits(v), bits(v2))
    assert bitexact > 0.4 * len(A)          # same FMA structure as OpenBLAS' kernels in most cases


def test_kmeans_vs_reference_faiss_golden(ob):
    g = golden("kmeans_ref.npz")
    for ci, (n, k, niter, max_samples, weighted, seed, plant) in enumerate(g["cases"]):
        x, w, cent = mg.km_inputs(int(n), int(k), bool(weighted), int(seed), bool(plant))
        mine = ob.kmeans_refine(x, w, int(n), cent, int(niter), int(max_samples)).astype(np.float32)
        ref = g["cent_%d" % ci]
        assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32)), "case %d" % ci


def test_subsample_prefix_matches_numpy_mt19937(ob):
    # faiss rand_perm(n, 1234): full Fisher-Yates with std::mt19937 raw outputs (random.cpp:184-194)
    n, take = 5000, 700
    mt = np.random.MT19937()
    mt._legacy_seeding(1234)
    raw = mt.random_raw(n)
    perm = np.arange(n)
    for i in range(n - 1):
        j = i + int(raw[i]) % (n - i)
        perm[i], perm[j] = perm[j], perm[i]
    out = np.zeros(take, dtype=np.int32)
    ob.lib().orc_kmeans_subsample_indices(n, take, 1234, out.ctypes.data_as(C.POINTER(C.c_int32)))
    assert np.array_equal(out, perm[:take])


def test_pipeline_regression_vectors(ob):
    """GQ/LQ/dither/end-to-end: oracle == its committed vectors (parity unpinned by a reference build)."""
    g = golden("pipeline_oracle.npz")
    assert np.all(g["probe_equal"] != 0)      # 1 = equal to the survey's shimmed build when generated, -1 = not run
    for ci, (w, h, K, cs, niter, dither, weighted, kind, seed) in enumerate(mg.PIPE_CASES):
        flat, wt = mg.pipe_input(w, h, kind, seed, weighted)
        ec, pal, pmap = ob.patolette(w, h, flat, wt, K, dither=dither, color_space=cs, kmeans_niter=niter,
                                     kmeans_max_samples=65536)
        assert ec == int(g["ec_%d" % ci])
        assert np.array_equal(np.asarray(pal), g["pal_%d" % ci], equal_nan=True), "case %d" % ci
        assert np.array_equal(pmap.astype(np.uint16), g["map_%d" % ci]), "case %d" % ci
