"""N > 1 path on CPU: world_size-2 gloo run of the batch sharding + final gather."""
import os
import subprocess
import sys

from patolette_amd import dist as pdist
from tests.util import ROOT


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
    assert out.read_text() == "OK"
