"""bench.py's multi-GPU contract: `python bench.py --gpus N` starts N RCCL ranks itself and reports the ranks
the group really had; the torch.distributed path on one rank measures the same thing as the plain path."""
import json
import os
import subprocess
import sys

import pytest

from tests.util import ROOT


def _bench(*flags, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--extra-streams", "0"] + list(flags),
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = json.loads(ln)
    return r, line


@pytest.mark.gpu
def test_rccl_path_on_one_rank_measures_what_the_plain_path_measures(gpu):
    r0, plain = _bench("--gpus", "1", "--steps", "12", "--warmup", "3")
    assert r0.returncode == 0 and plain is not None, r0.stdout[-2000:] + r0.stderr[-2000:]
    r1, dist = _bench("--gpus", "1", "--force-dist", "--steps", "12", "--warmup", "3")
    assert r1.returncode == 0 and dist is not None, r1.stdout[-2000:] + r1.stderr[-2000:]
    assert plain["n_gpus"] == 1 and dist["n_gpus"] == 1
    assert "RCCL gather" in dist["config"]["final_gather"] and plain["config"]["final_gather"].startswith("none")
    assert abs(dist["value"] / plain["value"] - 1.0) < 0.10, (plain["value"], dist["value"])
    assert dist["roofline"]["kernel"] == plain["roofline"]["kernel"]


@pytest.mark.gpu
def test_gpus_n_starts_n_ranks_or_fails_loudly(gpu):
    """`bench.py --gpus 2` with no launcher around it: two ranks on a box with two GPUs (n_gpus == 2 in the line), a non-zero
    exit and NO line on a one-GPU box -- never a one-GPU measurement labelled as something else."""
    have = gpu.patolette_amd_device_count()
    r, line = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", timeout=600)
    if have >= 2:
        assert r.returncode == 0 and line is not None and line["n_gpus"] == 2, r.stdout[-2000:] + r.stderr[-2000:]
        assert line["scaling"] == "weak"
    else:
        assert r.returncode != 0 and line is None, r.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("streams", [1, 2])
def test_n_greater_one_code_path_runs_on_ranks_sharing_a_device(gpu, streams):
    """bench.py's N > 1 branch (batch shard + final gather) EXECUTED on a one-GPU box: two ranks share the device
    (`--oversubscribe`: gloo + host staging for the collective, because RCCL refuses two ranks on one GPU); sharding by rank, the
    per-step asynchronous gather of the u8 maps, the palette gather, the index arithmetic of `map_ptr` and the max-over-ranks
    timing are the lines the RCCL run executes.  Rank 0 then quantises every rank's images itself, one call per image: what the
    gather delivered must be those maps and palettes bit for bit."""
    r, line = _bench("--gpus", "2", "--oversubscribe", "--check-gather", "--steps", "4", "--warmup", "2", "--config", "c2",
                     "--streams", str(streams), timeout=600)
    assert r.returncode == 0 and line is not None, r.stdout[-2000:] + r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert "TEST MODE" in line["config"]["final_gather"]
    gc = line["gather_check"]
    assert gc["ranks"] == 2 and gc["maps_compared"] == 2 * 4 * streams
    assert gc["map_mismatches"] == 0 and gc["palettes_identical"]
    # whole-job value = the pixels of BOTH ranks over the max-over-ranks time
    px = 1920 * 1080 * 4 * 2 * streams
    assert abs(line["value"] - px / (line["ms_per_step"] * 4e-3) / 1e6) <= 2e-3 * line["value"]
