"""The exact referee (tests/tie_prover.py) on the CPU: the oracle against ITSELF with its sequential sums taken back to front
(orc_set_sum_reversed) -- another member of the set of results that differ from the reference's only by the rounding of its sums,
which is what the HIP path's order-free sums are.  On degenerate content the two runs take different decisions (other cuts, other
palette rows); the referee must prove every decision of BOTH inside the rounding envelope of the exact optimum
(lib/src/quantize/local.c:102-177,256-307, global.c:189-298), and must turn red on deliberately wrong decision rules
(orc_set_fault).  The -m gpu tests apply the same referee to the HIP path's trace (tests/test_gpu_fuzz.py)."""
import numpy as np
import pytest

from tests import tie_prover as tp
from tests.util import scene


def content(rng, kind, h, w):
    n = h * w
    if kind == "noise":
        return rng.random((n, 3))
    if kind == "scene":
        return scene(h, w, int(rng.integers(0, 1000))).reshape(-1, 3)
    if kind == "post":                                            # posterised scene: 8 levels per channel
        return np.floor(scene(h, w, int(rng.integers(0, 1000))).reshape(-1, 3) * 8).clip(0, 7) / 7
    if kind == "few":
        pal = rng.random((int(rng.integers(1, 6)), 3))
        return pal[rng.integers(0, len(pal), size=n)]
    if kind == "flat":
        return np.tile(rng.random(3), (n, 1))
    if kind == "gradient":
        t = np.linspace(0, 1, n)[:, None]
        return np.clip(t * rng.random(3) + (1 - t) * rng.random(3), 0, 1)
    return rng.integers(0, 256, size=(n, 3)).astype(np.float64) / 255


def run(ob, data, wts, n, K, reversed_sums=0, fault=0):
    ob.set_sum_reversed(reversed_sums)
    ob.set_fault(fault)
    try:
        r = ob.quantize_clusters(data, wts, n, K, want_membership=False)
        return r, ob.last_split_trace()
    finally:
        ob.set_sum_reversed(0)
        ob.set_fault(0)


def cases(seed, count, kinds):
    rng = np.random.default_rng(seed)
    for _ in range(count):
        h, w = int(rng.integers(5, 60)), int(rng.integers(5, 60))
        kind = str(rng.choice(kinds))
        colors = np.ascontiguousarray(content(rng, kind, h, w))
        K = int(rng.choice([2, 3, 7, 16, 33, 64, 256]))
        cs = int(rng.integers(0, 3))
        wts = (1.0 + rng.random(h * w) * float(rng.choice([0.0, 3.0, 1000.0]))) if rng.integers(0, 2) else None
        yield kind, h * w, colors, wts, K, cs


@pytest.mark.parametrize("seed", [1, 2])
def test_both_summation_orders_are_inside_the_envelope(ob, seed):
    eig = tp.oracle_eigen(ob)
    diverged = {}
    ties = 0
    for kind, n, colors, wts, K, cs in cases(seed, 40, ["noise", "scene", "post", "few", "flat", "gradient", "u8"]):
        data = tp.converted(ob, ob.planar(colors), cs)
        _, ta = run(ob, data, wts, n, K)
        _, tb = run(ob, data, wts, n, K, reversed_sums=1)
        ref = tp.Referee(data, wts, n, eig)
        ra, rb = ref.check(ta, K), ref.check(tb, K)
        assert ra.ok, (kind, n, K, cs, ra.summary())
        assert rb.ok, (kind, n, K, cs, rb.summary())
        ties += len(ra.ties) + len(rb.ties)
        fd = tp.first_divergence(ta, tb)
        if fd:
            diverged[(kind, fd[0])] = diverged.get((kind, fd[0]), 0) + 1
    print("diverged:", diverged, "ties proven:", ties)
    assert diverged, "the reversed sums changed no decision: the test exercises nothing"
    assert all(k[0] in ("gradient", "post", "few", "flat", "u8") for k in diverged), diverged   # generic content: identical decisions


def test_generic_content_has_no_ties_and_one_answer(ob):
    """noise / photograph-like content: every decision is the exact optimum by a margin far beyond the envelope, in both orders"""
    eig = tp.oracle_eigen(ob)
    for kind, n, colors, wts, K, cs in cases(5, 12, ["noise", "scene"]):
        data = tp.converted(ob, ob.planar(colors), cs)
        ra_, ta = run(ob, data, wts, n, K)
        rb_, tb = run(ob, data, wts, n, K, reversed_sums=1)
        assert tp.first_divergence(ta, tb) is None
        assert np.allclose(ra_["centers"], rb_["centers"], rtol=0, atol=1e-12, equal_nan=True)
        rep = tp.Referee(data, wts, n, eig).check(ta, K)
        assert rep.ok, rep.summary()
        assert not [t for t in rep.ties if t["kind"] != "gq_cuts"], rep.ties   # (the DP check compares against an f64 optimum: gaps of 1e-16)


@pytest.mark.parametrize("fault,kinds", [(1, ["noise", "scene", "gradient"]), (2, ["noise", "scene", "gradient"])])
def test_a_wrong_decision_rule_turns_the_referee_red(ob, fault, kinds):
    """orc_set_fault(1): the cut one occupied bucket too far; (2): the greedy step takes the second best cluster"""
    eig = tp.oracle_eigen(ob)
    flagged = changed = 0
    for kind, n, colors, wts, K, cs in cases(7, 24, kinds):
        data = tp.converted(ob, ob.planar(colors), cs)
        _, ta = run(ob, data, wts, n, K)
        _, tb = run(ob, data, wts, n, K, fault=fault)
        if tp.first_divergence(ta, tb) is None:
            continue                                              # the fault had nothing to act on (no split committed, ...)
        changed += 1
        rep = tp.Referee(data, wts, n, eig).check(tb, K)
        flagged += (not rep.ok)
        assert not rep.ok, (kind, n, K, cs, "a wrong decision passed as a tie", rep.summary())
    assert changed >= 10 and flagged == changed


def test_the_last_maximum_is_a_member_of_the_tie_set(ob):
    """orc_set_fault(3): the cut at the LAST maximum of the objective instead of the first (vector.c:26-46).  Changes the result only
    where the objective is exactly tied in f64 -- inside the envelope by definition: the referee must stay green, and on generic
    content nothing changes at all."""
    eig = tp.oracle_eigen(ob)
    changed = 0
    for kind, n, colors, wts, K, cs in cases(9, 30, ["noise", "gradient", "few", "flat", "post"]):
        data = tp.converted(ob, ob.planar(colors), cs)
        _, ta = run(ob, data, wts, n, K)
        _, tb = run(ob, data, wts, n, K, fault=3)
        fd = tp.first_divergence(ta, tb)
        if fd:
            changed += 1
            assert kind != "noise", fd
        rep = tp.Referee(data, wts, n, eig).check(tb, K)
        assert rep.ok, (kind, n, K, cs, rep.summary())
    print("cases changed by the last-maximum rule:", changed)


def test_trace_replay_reproduces_the_oracles_own_result(ob):
    """orc_patolette_from_centers (patolette.c:246-336 behind given centres) fed with the oracle's own centres = orc_patolette"""
    rng = np.random.default_rng(3)
    for cs, dither, niter in ((0, False, 0), (1, True, 2), (2, False, 3), (2, True, 0)):
        w, h, K = 31, 23, 12
        colors = rng.random((w * h, 3))
        flat = ob.planar(colors)
        ec, pal, pmap = ob.patolette(w, h, flat, None, K, dither=dither, color_space=cs, kmeans_niter=niter)
        assert ec == 0
        r = ob.quantize_clusters(tp.converted(ob, flat, cs), None, w * h, K, want_membership=False)
        pal2, map2 = ob.patolette_from_centers(w, h, flat, None, K, r["centers"][:r["n_clusters"]], dither=dither, color_space=cs,
                                               kmeans_niter=niter)
        assert np.array_equal(pal, pal2) and np.array_equal(pmap, map2)
