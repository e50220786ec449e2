"""The image content bench.py TIMES beyond uniform noise (its `content` record: a smooth scene, the scene posterised to eight
levels per channel, noise with one colour over 30 % of the image), held to the CPU oracle at a size the oracle does in seconds.
Same generator (bench.content_images), same options as the headline configuration (256 colours, ICtCp + KMeans 32 it,
512^2 samples).  Follows lib/src/quantize/local.c:102-177 and global.c:189-298 through the oracle."""
import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def records(gpu, native, ob):
    cfg = bench.CONFIGS["c3"]
    w, h = bench.CONTENT_PARITY_SIZE
    out = {}
    for name, host in bench.content_images(w, h):
        out[name] = bench.content_parity(gpu, native, ob, cfg, host, w, h)
        print("content %s: %s" % (name, out[name]))
    return out


@pytest.mark.parametrize("name", ["scene", "dominant30"])
def test_photograph_like_content_is_the_oracles_result(records, name):
    r = records[name]
    assert r["palette_rows"] == r["palette_rows_oracle"] == 256
    assert r["palette_max_rel"] is not None and r["palette_max_rel"] <= 1e-9, r
    assert r["map_mismatches"] == 0, r


def test_posterised_content(records):
    """512 distinct colours at most (8 levels per channel): clusters of a few colours, where the reference's cut decisions hinge
    on the rounding of its sequential f64 sums (local.c:118-134).  Identical, or the difference is PROVEN tie noise: every decision
    of the HIP path's split trace within the rounding envelope of the exact optimum and the stages behind the quantisers reproduced
    by the oracle from the HIP path's centres (tests/tie_prover.py; bench.content_parity records the proof)."""
    r = records["posterised"]
    if r["verdict"].startswith("identical"):
        return
    assert r["tie_proof"]["proven"], r
