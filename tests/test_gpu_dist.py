"""BASELINE configs[4] geometry through the library's multi-GPU entry on the RCCL backend: `nccl` process group of
however many GPUs the box has (1 on the test box), 4096x4096 images, 256 colours + KMeans, results gathered from HBM.
32 images = one GPU's share of configs[4]'s 256 over 8 GPUs (12.9 GB of f64 pixels through the host)."""
import os
import subprocess
import sys

import pytest

from tests.util import ROOT


@pytest.mark.gpu
def test_sharded_batch_on_rccl_matches_per_image_calls(gpu, tmp_path):
    out = tmp_path / "result.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(ROOT, "tests", "dist_worker_nccl.py"), str(out), "4096", "32", "256"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    text = out.read_text()
    assert text.startswith("OK"), text
