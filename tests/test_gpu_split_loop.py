"""The split loop of the local quantiser (lib/src/quantize/local.c:318-404) driven from the DEVICE (pipeline.hip k_lq_control: the
children's eigen-solves, the greedy replay of local.c:347-390 and the next round's node list in one single-block kernel, no host turn
between rounds) against the host-driven loop: the same decisions -- split trace record for record -- the same centres bit for bit, the
same number of evaluations, and both equal to the oracle."""
import numpy as np
import pytest

from tests.test_tie_prover import content

pytestmark = pytest.mark.gpu


@pytest.fixture
def loop(gpu):
    yield lambda on_device: gpu.patolette_amd_set_split_loop(on_device)
    gpu.patolette_amd_set_split_loop(1)


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
        assert tr_h == tr_d, desc
        assert np.array_equal(cen_h, cen_d) and np.array_equal(pal_h, pal_d) and np.array_equal(map_h, map_d), desc
        for key in ("n_base_clusters", "n_clusters", "split_evals", "split_px", "lq_rounds"):
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
        assert tr_h == tr_d and np.array_equal(pal_h, pal_d) and np.array_equal(map_h, map_d)
        assert st_h["split_evals"] == st_d["split_evals"] and st_h["lq_rounds"] == st_d["lq_rounds"]
        ec, pal_o, map_o = ob.patolette(w, h, flat, wts, 256, dither=False, color_space=cs, kmeans_niter=0)
        assert ec == 0 and np.allclose(pal_d, pal_o, rtol=0, atol=1e-9) and np.array_equal(map_d, map_o)
        print("%dx%d: %d rounds, %d evaluations; ms_lq host loop %.3f, device loop %.3f" %
              (w, h, st_d["lq_rounds"], st_d["split_evals"], st_h["ms_lq"], st_d["ms_lq"]))
