"""SURVEY.md 8(f)-4 / 8(e) last row: ONE image dealt out over several GPUs (`patolette_amd_slice`).

The test box has one GPU, so the group is two processes sharing it, exchanging the per-node reductions over gloo (the library
stages its buffers through pinned host memory); the RCCL form of the same entry (device buffers, `nccl`) runs with the
group the box offers (one rank).  Each rank compares its results bit for bit with the whole image quantised on one GPU."""
import os
import subprocess
import sys

import pytest

from tests.util import ROOT


def _run(tmp_path, backend, nproc, port):
    out = tmp_path / "slice"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "slice_worker.py"), str(out), backend]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for rank in range(nproc):
        text = open("%s.%d" % (out, rank)).read()
        assert text.startswith("OK"), text
    return text


@pytest.mark.gpu
def test_image_sliced_over_two_ranks_equals_one_gpu(gpu, tmp_path):
    _run(tmp_path, "gloo", 2, 29651)


@pytest.mark.gpu
def test_image_sliced_over_three_ranks_equals_one_gpu(gpu, tmp_path):
    _run(tmp_path, "gloo", 3, 29652)


@pytest.mark.gpu
def test_sliced_entry_on_rccl_device_buffers(gpu, tmp_path):
    _run(tmp_path, "nccl", 1, 29653)
