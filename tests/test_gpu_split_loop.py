"""The split loop of the local quantiser (lib/src/quantize/local.c:318-404) driven from the DEVICE (pipeline.hip k_lq_children /
k_lq_select: the children's moments, bounds and eigen-solves and the next round's node list made on the device, round after round
without a host turn; the greedy loop of local.c:347-390 replayed once at the end) against the host-driven loop: the same decisions --
split trace record for record -- the same centres, palette and map bit for bit, and both equal to the oracle.  Only the SET of
evaluated candidate nodes may differ (the device's selection rule is exact-safe, the host's a heuristic checked in lock-step)."""
import numpy as np
import pytest

from tests.test_tie_prover import content

pytestmark = pytest.mark.gpu


@pytest.fixture
def loop(gpu):
    yield lambda on_device: gpu.patolette_amd_set_split_loop(on_device)
    gpu.patolette_amd_set_split_loop(1)


def _same_decisions(ta, tb):
    """Two split traces take the same decisions: header and every record equal, the moments (distortions, covariance, hence
    benefit and axis) to 1e-12 relative -- the children's centred sums are taken per run of tiles before they are split onto the
    exact grids, so their last bits move with the round's tile list, which the two drivers compose differently."""
    if any(ta[k] != tb[k] for k in ta if k not in ("splits", "gq_axis", "gq_cov6")) or len(ta["splits"]) != len(tb["splits"]):
        return False
    for a, b in zip(ta["splits"], tb["splits"]):
        for k in ("row", "new_row", "split", "degenerate", "n", "n_left", "n_right", "sw"):
            if a[k] != b[k]:
                return False
        for k in ("dist", "dist_left", "dist_right", "benefit"):
            if abs(a[k] - b[k]) > 1e-12 * max(abs(a["dist"]), 1e-300):
                return False
        if not np.allclose(a["axis"], b["axis"], rtol=0, atol=1e-9) or not np.allclose(a["cov6"], b["cov6"], rtol=1e-10, atol=1e-300):
            return False
    return True


def _run(p, native, w, h, colors, K, cs, wts):
    ok, pal, pmap, msg = p.quantize(w, h, colors, K, dither=False, color_space=cs, tile_size=0, kmeans_niter=0, weights=wts)
    assert ok, msg
    return pal, pmap, native.last_split_trace(), native.last_cluster_centers(), p.last_stats()


@pytest.mark.parametrize("seed", [1, 2])
def test_device_driven_loop_equals_host_driven_loop(gpu, native, ob, loop, seed):
    import patolette_amd as p
    rng = np.random.default_rng(seed)
    rounds = []
    for case in range(30):
        h, w = int(rng.integers(3, 200)), int(rng.integers(3, 200))
        kind = str(rng.choice(["noise", "scene", "post", "few", "flat", "gradient", "u8"]))
        if kind in ("scene", "post") and (h <= 4 or w <= 4):
            kind = "noise"
        colors = np.ascontiguousarray(content(rng, kind, h, w))
        K = int(rng.choice([2, 3, 7, 16, 33, 64, 200, 256]))
        cs = int(rng.integers(0, 3))
        wts = (1.0 + rng.random(h * w) * float(rng.choice([0.0, 3.0, 1000.0]))) if rng.integers(0, 2) else None
        loop(0)
        pal_h, map_h, tr_h, cen_h, st_h = _run(p, native, w, h, colors, K, cs, wts)
        loop(1)
        pal_d, map_d, tr_d, cen_d, st_d = _run(p, native, w, h, colors, K, cs, wts)
        desc = (seed, case, w, h, kind, K, cs, wts is not None)
        assert _same_decisions(tr_h, tr_d), desc
        assert np.array_equal(cen_h, cen_d) and np.array_equal(pal_h, pal_d) and np.array_equal(map_h, map_d), desc
        for key in ("n_base_clusters", "n_clusters"):
            assert st_h[key] == st_d[key], (desc, key, st_h[key], st_d[key])
        rounds.append(st_d["lq_rounds"])
    assert max(rounds) >= 3


def test_device_driven_loop_at_bench_sizes(gpu, native, ob, loop):
    """1920x1080 and 2048x2048 noise, K = 256 (BASELINE configs[1] and a weighted CIELuv image): device loop == host loop == oracle"""
    import patolette_amd as p
    for (w, h, cs, weighted) in ((1920, 1080, 2, False), (2048, 2048, 1, True)):
        n = w * h
        flat = ob.image(n, 3)
        colors = flat.reshape(3, n).T.copy()
        wts = ob.weights(n, 3) if weighted else None
        loop(0)
        pal_h, map_h, tr_h, cen_h, st_h = _run(p, native, w, h, colors, 256, cs, wts)
        loop(1)
        pal_d, map_d, tr_d, cen_d, st_d = _run(p, native, w, h, colors, 256, cs, wts)
        assert _same_decisions(tr_h, tr_d) and np.array_equal(pal_h, pal_d) and np.array_equal(map_h, map_d)
        ec, pal_o, map_o = ob.patolette(w, h, flat, wts, 256, dither=False, color_space=cs, kmeans_niter=0)
        assert ec == 0 and np.allclose(pal_d, pal_o, rtol=0, atol=1e-9) and np.array_equal(map_d, map_o)
        print("%dx%d: host loop %d rounds, %d evaluations, ms_lq %.3f; device loop %d rounds, %d evaluations, ms_lq %.3f" %
              (w, h, st_h["lq_rounds"], st_h["split_evals"], st_h["ms_lq"], st_d["lq_rounds"], st_d["split_evals"], st_d["ms_lq"]))


@pytest.fixture
def gq(gpu):
    yield lambda on_device: gpu.patolette_amd_set_global_quantiser(on_device)
    gpu.patolette_amd_set_global_quantiser(1)


def _blobs(rng, n, nblobs, spread):
    """`nblobs` tight clouds strung along one direction of the colour cube: the global quantiser (global.c:189-298) finds several cells
    biased along the principal axis and goes on beyond two base clusters"""
    direction = rng.random(3) + 0.2
    direction /= np.linalg.norm(direction)
    centres = 0.15 + 0.7 * np.sort(rng.random(nblobs))[:, None] * direction[None, :] / direction.max()
    which = rng.integers(0, nblobs, size=n)
    return np.clip(centres[which] + spread * rng.standard_normal((n, 3)), 0.0, 1.0)


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_global_quantiser_on_the_device_equals_the_hosts_turn(gpu, native, ob, loop, gq, seed):
    """k_gq_control (prefix chains, bias termination test, the dynamic programme's full steps and backtrack, bucket -> base cluster
    table, base clusters' records -- all on the device) against the host's turn over the same downloaded table, and against the oracle:
    same number of base clusters and cuts (split trace header), same centres, palette and map.  Content with MORE THAN TWO base
    clusters included (the DP's full steps run only then), both split loops, weighted and unweighted, several palette sizes."""
    import patolette_amd as p
    rng = np.random.default_rng(seed)
    seen = []
    for case in range(14):
        h, w = int(rng.integers(40, 160)), int(rng.integers(40, 160))
        n = h * w
        kind = case % 4
        if kind == 0:
            colors = _blobs(rng, n, int(rng.integers(3, 9)), float(rng.choice([0.002, 0.01, 0.03])))
        elif kind == 1:
            colors = np.ascontiguousarray(content(rng, "scene", h, w))
        elif kind == 2:
            colors = _blobs(rng, n, int(rng.integers(2, 5)), 0.05)
        else:
            colors = np.ascontiguousarray(content(rng, str(rng.choice(["noise", "post", "gradient"])), h, w))
        colors = np.ascontiguousarray(colors)
        K = int(rng.choice([13, 16, 40, 256]))
        cs = int(rng.integers(0, 3))
        wts = (1.0 + 3.0 * rng.random(n)) if rng.integers(0, 2) else None
        on_device_loop = int(rng.integers(0, 2))
        loop(on_device_loop)
        gq(0)
        pal_h, map_h, tr_h, cen_h, st_h = _run(p, native, w, h, colors, K, cs, wts)
        gq(1)
        pal_d, map_d, tr_d, cen_d, st_d = _run(p, native, w, h, colors, K, cs, wts)
        desc = (seed, case, w, h, kind, K, cs, wts is not None, on_device_loop, st_h["n_base_clusters"])
        assert st_h["n_base_clusters"] == st_d["n_base_clusters"] and tr_h["n_base"] == tr_d["n_base"], desc
        assert list(tr_h["gq_cuts"]) == list(tr_d["gq_cuts"]), desc
        assert _same_decisions(tr_h, tr_d), desc
        assert np.array_equal(cen_h, cen_d) and np.array_equal(pal_h, pal_d) and np.array_equal(map_h, map_d), desc
        ec, pal_o, map_o = ob.patolette(w, h, ob.planar(colors), wts, K, dither=False, color_space=cs, kmeans_niter=0)
        assert ec == 0, desc
        if kind in (0, 1, 2) or np.allclose(pal_d, pal_o, rtol=0, atol=1e-9):
            assert np.allclose(pal_d, pal_o, rtol=0, atol=1e-9) and np.array_equal(map_d, map_o), desc
        seen.append(st_d["n_base_clusters"])
    print("seed %d: base clusters per case %s" % (seed, seen))
    assert max(seen) >= 4 and sum(1 for k in seen if k >= 3) >= 3, seen
