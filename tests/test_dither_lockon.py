"""The property the segment-parallel dither (map.hip, k_dither_seg) rests on, measured with the CPU oracle alone.

The reference pushes `original pixel - chosen palette colour` into its 16-entry error queue
(lib/src/dither/riemersma.c:333-340), so the whole state of the chain is a pure function of the last sixteen
(pixel, chosen index) pairs.  A chain started from a ZERO queue anywhere on the Hilbert curve is therefore
bit-identical to the true chain from the moment it has made the same choice sixteen times in a row -- which is
checkable after the fact.  This file measures how long that takes (the warm-up a speculative segment needs) and
asserts the 'identical thereafter' part, on the content classes the GPU path is timed on.

Method: on a 2^L square every curve position is in the image, so rotating the pixel sequence along
`hilbert_order` by s makes the oracle's own chain (which always starts from zeros at curve position 0) start
from zeros at original curve position s.
"""
import numpy as np
import pytest

from tests import util

SIDE = 256
N = SIDE * SIDE
RUN = 16384        # steps each speculative chain is followed for


def _palette_and_pixels(ob, flat, K):
    """Palette + pixels in the space the dither runs in (linear Rec2020), as patolette.c:268-299 hands them over."""
    ec, pal, _ = ob.patolette(SIDE, SIDE, flat, None, K, dither=False, color_space=2, kmeans_niter=0)
    assert ec == 0
    pal = pal[pal[:, 0] >= 0]
    pal2020 = ob.convert("srgb_to_rec2020", ob.planar(pal).copy()).reshape(3, -1).T.copy()
    px2020 = ob.convert("srgb_to_rec2020", flat.copy())
    return pal2020, px2020


def _images(ob):
    noise = ob.image(N, 3)
    sc = util.scene(SIDE, SIDE, 1)
    scene = np.concatenate([sc[:, :, c].reshape(-1) for c in range(3)])
    flat_grey = np.clip(0.5 + 0.01 * (ob.image(N, 9) - 0.5), 0, 1)          # near-flat: tiny errors, choices flip slowly
    return {"noise": noise, "scene": scene, "nearflat": flat_grey}


def _lock_step(ob, px, order, pal, true_map, s, run):
    """Steps a zero-queue chain started at curve position s needs until 16 consecutive choices equal the true chain's;
    asserts every later choice (to the end of the run) is identical."""
    run = min(run, N - s)
    rot = np.empty(3 * N)
    src = np.concatenate([order[s:], order[:s]])                              # curve position i of the rotated image = position s + i
    for c in range(3):
        rot[c * N:(c + 1) * N][order] = px[c * N:(c + 1) * N][src]
    got = ob.dither_prefix(rot, SIDE, SIDE, pal, run)
    mine = got[order[:run]].astype(np.int64)
    ref = true_map[order[s:s + run]].astype(np.int64)
    same = mine == ref
    streak = 0
    for i in range(run):
        streak = streak + 1 if same[i] else 0
        if streak == 16:
            assert same[i:].all(), "a chain that made the true chain's last 16 choices must stay on it"
            return i + 1
    return None


@pytest.mark.parametrize("K", [16, 256])
def test_zero_queue_chain_locks_onto_the_true_chain(ob, K):
    rng = np.random.default_rng(5)
    order = ob.hilbert_order(SIDE, SIDE).astype(np.int64)
    assert order.size == N
    worst = {}
    for name, flat in _images(ob).items():
        pal, px = _palette_and_pixels(ob, flat, K)
        true_map = ob.dither(px, SIDE, SIDE, pal)
        locks = []
        for s in rng.integers(1, N - 20000, size=12):
            lk = _lock_step(ob, px, order, pal, true_map, int(s), RUN)
            assert lk is not None, "%s K=%d start %d: no lock within %d steps" % (name, K, s, RUN)
            locks.append(lk)
        worst[name] = max(locks)
        assert min(locks) >= 16
    print("dither lock-on, K=%d: worst steps per content %s" % (K, worst))
    # the product's default warm-up (launch_dither, map.hip: 1024 pixels) is sized from these; a miss is repaired, never wrong
    assert max(worst.values()) <= 1024


def _d2xy(L, d):
    """Curve position -> (x, y); the walk of traverse_level(L, UP) from (0, 0) (riemersma.c:176-257), checked below against the
    oracle's recorded walk."""
    x = y = 0
    t = d
    for lv in range(L):
        s = 1 << lv
        rx = 1 & (t >> 1)
        ry = 1 & (t ^ rx)
        if ry == 0:
            if rx == 1:
                x, y = s - 1 - x, s - 1 - y
            x, y = y, x
        x += s * rx
        y += s * ry
        t >>= 2
    return x, y


@pytest.mark.parametrize("w,h", [(64, 64), (100, 37), (37, 100), (130, 129), (9, 300), (257, 16)])
def test_run_boundaries_of_the_segment_parallel_dither(ob, native, w, h):
    """dither_locate (map.hip; the host copy of the function the kernel cuts the curve with) against the oracle's walk."""
    import ctypes as C
    L = 0
    while (1 << L) < max(w, h):
        L += 1
    order = ob.hilbert_order(w, h).astype(np.int64)
    assert order.size == w * h
    pos = []                                                      # curve position of every in-image pixel, in curve order
    for d in range(1 << (2 * L)):
        x, y = _d2xy(L, d)
        if x < w and y < h:
            pos.append(d)
            assert order[len(pos) - 1] == y * w + x
    pos = np.array(pos)
    lib = native.lib()
    d, c = C.c_ulonglong(), C.c_ulonglong()
    rng = np.random.default_rng(w * 1000 + h)
    for t in list(rng.integers(0, w * h, size=200)) + [0, w * h - 1]:
        lib.patolette_amd_debug_dither_locate(w, h, int(t), C.byref(d), C.byref(c))
        assert d.value % 64 == 0 and d.value <= pos[t] < d.value + 64
        assert c.value == np.searchsorted(pos, d.value)
    lib.patolette_amd_debug_dither_locate(w, h, w * h, C.byref(d), C.byref(c))
    assert d.value == 1 << (2 * L) and c.value == w * h


def test_the_queue_rotations_header_is_what_its_generator_writes():
    """patolette_amd/csrc/dither_slots.h (the sixteen rotations of the lane kernel's error queue) is generated: the committed file
    must be the generator's output, and every rotation must read the slots in the order j, j + 1, ... with the weights 0 .. 14."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_dither_slots.py")], capture_output=True, text=True, check=True).stdout
    have = open(os.path.join(root, "patolette_amd", "csrc", "dither_slots.h")).read()
    body = have[have.index("// ---- GENERATED"):]
    want = gen.replace("    // ---- GENERATED", "// ---- GENERATED").replace("    // ---- end", "// ---- end")
    assert body == want
    for J in range(16):
        m = re.search(r"case %d: (e0 \+= .*?) \\\n" % J, body)
        terms = re.findall(r"e0 \+= q0_(\d+) \* wts\.w\[(\d+)\];", m.group(1))
        assert [(int(a), int(b)) for a, b in terms] == [((J + i) & 15, i) for i in range(15)]


@pytest.mark.parametrize("k", [4, 16, 64, 256])
def test_on_one_colour_a_recurrence_of_sixteen_choices_gives_the_rest_of_the_chain(ob, k):
    """What the lane dither's walk through FLAT stretches rests on (map.hip, k_dither_lane_repair `jump`): on pixels of one colour the
    chain's state is its last sixteen choices, so once choices[T-16:T] == choices[T-16-P:T-P] the chain repeats with period P for
    as long as the colour lasts.  With the oracle's chain over a flat image: the first such (T, P) predicts every later choice."""
    rng = np.random.default_rng(100 + k)
    w = h = 128
    n = w * h
    order = np.asarray(ob.hilbert_order(w, h))
    for trial in range(3):
        pal = rng.random((k, 3))
        c = rng.random(3)
        flat = np.concatenate([np.full(n, c[j]) for j in range(3)])
        seq = np.asarray(ob.dither(flat, w, h, pal))[order]
        found = None
        for T in range(48, 4096, 16):
            win = seq[T - 16:T]
            for P in range(1, T - 16 + 1):
                if np.array_equal(seq[T - 16 - P:T - P], win):
                    found = (T, P)
                    break
            if found:
                break
        assert found, "no recurrence within 4096 steps (k = %d)" % k
        T, P = found
        pred = seq[T - P + (np.arange(T, n) - T) % P]
        assert np.array_equal(pred, seq[T:]), (k, T, P)
