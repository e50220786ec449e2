"""Randomised end-to-end sweep of the HIP path against the oracle (small images, every option combination).

The requirement for every case: the result is the oracle's -- palette within 1e-9, map bit for bit -- or the difference is PROVEN
tie noise by tests/tie_prover.py: every decision the HIP path's quantisers took (its split trace) is the exact optimum or within
the rounding envelope of the reference's own sequential sums (local.c:102-177, 256-307, global.c:189-298; exact integer
arithmetic on the cluster's pixels), the same holds for the oracle's trace, and everything behind the quantisers -- KMeans,
mapping / dithering, write-out -- replayed by the oracle from the HIP path's cluster centres gives the HIP path's palette and
map.  Degenerate content (perfect gradients, posterised scenes, a handful of colours, one flat colour) is where the reference's
decisions are ties in exact arithmetic: another summation order gives other palette rows and another image.
Generic content (noise, photograph-like scenes, 8-bit noise) must agree exactly."""
import numpy as np
import pytest

from tests import tie_prover as tp
from tests.test_tie_prover import content

pytestmark = pytest.mark.gpu

KINDS = ["noise", "scene", "few", "flat", "u8", "gradient", "post"]


def _case(rng):
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    n = h * w
    kind = str(rng.choice(KINDS))
    if kind in ("scene", "post") and (h <= 4 or w <= 4):
        kind = "noise"
    colors = content(rng, kind, h, w)
    opts = dict(K=int(rng.choice([1, 2, 3, 7, 16, 33, 64, 256, 300])), cs=int(rng.integers(0, 3)), dither=bool(rng.integers(0, 2)),
                niter=int(rng.choice([0, 0, 1, 3])))
    wts = (1.0 + rng.random(n) * float(rng.choice([0.0, 3.0, 1000.0]))) if rng.integers(0, 2) else None
    return w, h, kind, np.ascontiguousarray(colors), wts, opts


def _run_case(p, ob, native, gpu, w, h, colors, wts, o):
    """-> None if the HIP path's result is the oracle's, else the prover's explanation (raises if there is none)"""
    ok, pal_g, map_g, _ = p.quantize(w, h, colors, o["K"], dither=o["dither"], color_space=o["cs"], tile_size=0,
                                     kmeans_niter=o["niter"], kmeans_max_samples=512 ** 2, weights=wts)
    ec, pal_o, map_o = ob.patolette(w, h, ob.planar(colors), wts, o["K"], dither=o["dither"], color_space=o["cs"],
                                    kmeans_niter=o["niter"], kmeans_max_samples=512 ** 2)
    assert ok == (ec == 0)
    if not ok:
        return None
    if np.allclose(pal_g, pal_o, rtol=0, atol=1e-9) and np.array_equal(map_g, map_o):
        return None
    return tp.explain_divergence(ob, native, gpu, w, h, ob.planar(colors), wts, o["K"], o["cs"], o["dither"], o["niter"], 512 ** 2, pal_g, map_g)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_configurations_match_the_oracle_or_are_proven_ties(gpu, native, ob, seed):
    import patolette_amd as p
    rng = np.random.default_rng(seed)
    exact = 0
    proven = {}
    for case in range(60):
        w, h, kind, colors, wts, o = _case(rng)
        try:
            why = _run_case(p, ob, native, gpu, w, h, colors, wts, o)
        except AssertionError as e:
            raise AssertionError("seed %d case %d %dx%d %s %r weighted=%s: %s" % (seed, case, w, h, kind, o, wts is not None, e))
        if why is None:
            exact += 1
            continue
        assert kind not in ("noise", "scene"), ("generic content must agree exactly", seed, case, kind, o, why)
        key = (kind, why["first"][0] if why["first"] else "conversion")
        proven[key] = proven.get(key, 0) + 1
    print("seed %d: exact %d, proven ties by (content, first differing decision): %s" % (seed, exact, proven))
    assert exact >= 30, (exact, proven)


@pytest.fixture
def fault(gpu):
    yield lambda which: gpu.patolette_amd_debug_fault(which)
    gpu.patolette_amd_debug_fault(0)


@pytest.mark.parametrize("which", [1, 2])
def test_a_deliberately_broken_build_turns_the_prover_red(gpu, native, ob, fault, which):
    """patolette_amd_debug_fault(1): k_cut takes one occupied bucket too many; (2): the greedy replay commits the second best cluster.
    Either must be reported as a decision outside the envelope, not as tie noise.  (On a perfect gradient a wrong rule can land on
    the OTHER member of an exact tie -- an odd run of evenly spaced colours cut 59 | 60 or 60 | 59 -- which is then rightly a tie.)"""
    import patolette_amd as p
    rng = np.random.default_rng(100 + which)
    changed = red = 0
    for case in range(24):
        h, w = int(rng.integers(8, 60)), int(rng.integers(8, 60))
        kind = ["noise", "gradient", "scene"][case % 3]
        colors = np.ascontiguousarray(content(rng, kind, h, w))
        o = dict(K=int(rng.choice([7, 16, 64])), cs=int(rng.integers(0, 3)), dither=False, niter=0)
        fault(which)
        try:
            why = _run_case(p, ob, native, gpu, w, h, colors, None, o)
        except AssertionError as e:
            assert "outside the rounding envelope" in str(e), e
            red += 1
            changed += 1
            continue
        finally:
            fault(0)
        changed += why is not None
        assert why is None or kind == "gradient", ("a wrong decision rule passed as tie noise", case, kind, o, why)
    assert changed >= 12 and red >= 12, (changed, red)


def test_the_last_maximum_rule_stays_green(gpu, native, ob, fault):
    """patolette_amd_debug_fault(3): k_cut takes the LAST maximum of the objective: identical on generic content, a proven tie elsewhere"""
    import patolette_amd as p
    rng = np.random.default_rng(77)
    for case in range(18):
        h, w = int(rng.integers(8, 60)), int(rng.integers(8, 60))
        kind = ["noise", "gradient", "few"][case % 3]
        colors = np.ascontiguousarray(content(rng, kind, h, w))
        o = dict(K=int(rng.choice([7, 16, 64])), cs=int(rng.integers(0, 3)), dither=False, niter=0)
        fault(3)
        try:
            why = _run_case(p, ob, native, gpu, w, h, colors, None, o)
        finally:
            fault(0)
        if kind == "noise":
            assert why is None, why
