"""bench.py's command line on a box WITHOUT a GPU (the driver's contract): the HIP path has no CPU fallback, so
every way of starting it must fail loudly -- and `--gpus N` must never quietly measure fewer ranks than it was asked for."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    if env:
        e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)


def _no_gpu(native):
    return native.lib().patolette_amd_device_count() <= 0


def test_one_gpu_without_a_device_fails_loudly(native):
    if not _no_gpu(native):
        pytest.skip("a HIP device is visible")
    r = _run("--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras")
    assert r.returncode != 0
    assert "no HIP device" in (r.stdout + r.stderr)
    assert "\"metric\"" not in r.stdout                       # no line, not even a partial one


def test_more_gpus_than_devices_is_refused_before_any_rank_starts(native):
    if not _no_gpu(native):
        pytest.skip("a HIP device is visible")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "--gpus 2" in (r.stdout + r.stderr) and "HIP device" in (r.stdout + r.stderr)


def test_world_size_that_differs_from_gpus_is_refused():
    """Under an external launcher (WORLD_SIZE set) the line must describe the ranks that exist: `--gpus 4` inside a 2-rank launch is
    a contradiction, not a 4-GPU measurement."""
    r = _run("--gpus", "4", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stdout + r.stderr)


def test_unknown_configuration_is_an_argument_error():
    r = _run("--config", "nope")
    assert r.returncode == 2 and "invalid choice" in r.stderr
