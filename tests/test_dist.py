"""N > 1 paths on CPU: world_size-2 gloo run of the batch sharding + final gather, and the exchange protocol of the
within-image sharding (tests/split_model.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from patolette_amd import dist as pdist
from tests import split_model as psplit
from tests.util import ROOT


def _node(seed, n, weighted):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 3)) * np.array([0.15, 0.11, 0.15]) + np.array([0.005, -0.056, -0.05])      # ICtCp-like box
    w = (1.0 + 3.0 * rng.random(n) ** 3) if weighted else None
    axis = rng.standard_normal(3)
    axis /= np.linalg.norm(axis)
    P = int(np.ceil(np.log2(max(n, 2))))
    return c, w, axis, psplit.make_bink(2 if weighted else 0, P)


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("cuts", [(0.5,), (0.1, 0.55), (0.0, 0.3, 0.31)])
def test_within_image_split_is_exact_for_any_dealing_of_the_pixels(weighted, cuts):
    """The reduced moment table of a node -- and so the cut every rank takes -- does not depend on how the pixels are dealt
    out over the ranks: two-part binned sums are exact in any order (devutil.h), extrema go through ordered integer keys."""
    n = 50000
    c, w, axis, bink = _node(3, n, weighted)
    whole_cut, whole_b, whole_tab = psplit.split_node_sharded(c, w, axis, bink, lambda a, op: a)
    bounds = [0] + [int(f * n) for f in cuts] + [n]
    parts = [(c[a:b], None if w is None else w[a:b]) for a, b in zip(bounds, bounds[1:])]
    world = len(parts)
    # lock-step "ranks" in one process: every reduction sees all ranks' contributions
    ext = [psplit.local_extrema(pc, axis) for pc, _ in parts]
    mn = psplit.key_f64(np.minimum.reduce([e[:1] for e in ext]))[0]
    mx = psplit.key_f64(np.maximum.reduce([e[1:] for e in ext]))[0]
    tabs = [psplit.local_tables(pc, pw, axis, mn, mx, bink)[0] for pc, pw in parts]
    red = dict(parts=np.add.reduce([t["parts"] for t in tabs]), count=np.add.reduce([t["count"] for t in tabs]),
               size=np.add.reduce([t["size"] for t in tabs]))
    assert world >= 2
    assert np.array_equal(red["parts"].view(np.uint64), whole_tab["parts"].view(np.uint64))       # bit for bit
    assert np.array_equal(red["count"], whole_tab["count"]) and np.array_equal(red["size"], whole_tab["size"])
    assert psplit.cut_of(red, weighted) == whole_cut
    # the same table when the pixels of every rank arrive in another order (atomics on a GPU add in any order)
    perm = np.random.default_rng(9).permutation(n)
    _, _, shuffled = psplit.split_node_sharded(c[perm], None if w is None else w[perm], axis, bink, lambda a, op: a)
    assert np.array_equal(shuffled["parts"].view(np.uint64), whole_tab["parts"].view(np.uint64))


def test_shard_partitions_batch():
    for count in (0, 1, 5, 8, 256, 257):
        for world in (1, 2, 3, 8):
            blocks = [pdist.shard(count, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(n for _, n in blocks) == count
            for (s0, n0), (s1, _) in zip(blocks, blocks[1:]):
                assert s0 + n0 == s1
            assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1


def test_two_rank_gloo_batch(tmp_path):
    out = tmp_path / "result.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(ROOT, "tests", "dist_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert out.read_text() == "OK"                            # batch sharding and the within-image exchange, both
