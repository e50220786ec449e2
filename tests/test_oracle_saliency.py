"""CPU checks of the saliency oracle (oracle/saliency.py, orc_mbd): no GPU.

The reference's binding cannot be imported here (skimage absent), so these are known-answer and
cross-formulation checks, not a pinning against the reference's own output (oracle/saliency.py header).
"""
import numpy as np
import pytest

from tests.util import scene


def test_rgb2lab_known_answers():
    from oracle import saliency
    # published CIELAB (D65, 2 deg) of the sRGB primaries / white / mid grey
    table = {(1, 0, 0): (53.2408, 80.0925, 67.2032), (0, 1, 0): (87.7347, -86.1827, 83.1793),
             (0, 0, 1): (32.2970, 79.1875, -107.8602), (1, 1, 1): (100.0, 0.0, 0.0), (0, 0, 0): (0.0, 0.0, 0.0),
             (0.5, 0.5, 0.5): (53.3890, 0.0, 0.0)}
    rgb = np.array(list(table.keys()), dtype=np.float64).reshape(1, -1, 3)
    lab = saliency.rgb2lab(rgb).reshape(-1, 3)
    assert np.allclose(lab, np.array(list(table.values())), atol=0.02)


def _mbd_python(img, iters):
    """Direct transcription of the definition (dict-free loops over explicit neighbour offsets), for tiny images."""
    rows, cols = img.shape
    f = np.float32
    L, U = img.astype(f).copy(), img.astype(f).copy()
    D = np.full((rows, cols), np.inf, dtype=f)
    D[0, :] = D[-1, :] = 0
    D[:, 0] = D[:, -1] = 0
    for p in range(iters):
        forward = p % 2 == 1
        xs = range(1, rows - 1) if forward else range(rows - 2, 1, -1)
        ys = list(range(1, cols - 1) if forward else range(cols - 2, 1, -1))
        s = -1 if forward else 1
        for x in xs:
            for y in ys:
                ix, d = img[x, y], D[x, y]
                cands = []
                for (nx, ny) in ((x + s, y), (x, y + s)):
                    hi, lo = max(U[nx, ny], ix), min(L[nx, ny], ix)
                    cands.append((f(hi - lo), hi, lo))
                (b1, h1, l1), (b2, h2, l2) = cands
                if d <= b1 and d <= b2:
                    continue
                if b1 < d and b1 <= b2:
                    D[x, y], U[x, y], L[x, y] = b1, h1, l1
                else:
                    D[x, y], U[x, y], L[x, y] = b2, h2, l2
    return D


@pytest.mark.parametrize("rows,cols", [(4, 4), (5, 9), (12, 7), (17, 23)])
def test_mbd_c_matches_python_definition(ob, rows, cols):
    rng = np.random.default_rng(rows * 100 + cols)
    img = rng.random((rows, cols), dtype=np.float32)
    for iters in (1, 2, 3, 4):
        want = _mbd_python(img, iters)
        got = ob.mbd(img, iters)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert ob.mbd(np.zeros((3, 9), dtype=np.float32)) is None


def test_get_weights_properties():
    from oracle import saliency
    img = scene(90, 130, 4)
    w512 = saliency.get_weights(img, 512.0)
    w64 = saliency.get_weights(img, 64.0)
    assert w512.shape == (90 * 130,) and np.all(np.isfinite(w512)) and np.all(w512 >= 1.0)
    # weight - 1 scales with 1 / tile_size^2 (patolette.pyx:313)
    assert np.allclose((w64 - 1.0), (w512 - 1.0) * 64.0, rtol=1e-12)
    # the sigmoid keeps sal in (0, 1): weights bounded by 1 + N / tile^2
    assert w64.max() <= 1.0 + 90 * 130 / 64.0 ** 2
    assert saliency.check_shape(3, 500) and saliency.check_shape(9, 9) and saliency.check_shape(6, 2000)
    assert saliency.check_shape(40, 40) is None
    with pytest.raises(np.linalg.LinAlgError):
        saliency.get_weights(np.full((40, 40, 3), 0.25), 512.0)
