"""Host-driven vs device-driven split loop on one case: first differing trace record, field by field (GPU box)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
from patolette_amd import _native as native
from tests.test_tie_prover import content
L = native.lib()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed)
for case in range(ncase):
    h, w = int(rng.integers(3, 200)), int(rng.integers(3, 200))
    kind = str(rng.choice(["noise", "scene", "post", "few", "flat", "gradient", "u8"]))
    if kind in ("scene", "post") and (h <= 4 or w <= 4):
        kind = "noise"
    colors = np.ascontiguousarray(content(rng, kind, h, w))
    K = int(rng.choice([2, 3, 7, 16, 33, 64, 200, 256]))
    cs = int(rng.integers(0, 3))
    wts = (1.0 + rng.random(h * w) * float(rng.choice([0.0, 3.0, 1000.0]))) if rng.integers(0, 2) else None
    res = []
    for mode in (0, 1):
        L.patolette_amd_set_split_loop(mode)
        ok, pal, pmap, msg = p.quantize(w, h, colors, K, dither=False, color_space=cs, tile_size=0, kmeans_niter=0, weights=wts)
        res.append((native.last_split_trace(), p.last_stats(), pal))
    L.patolette_amd_set_split_loop(1)
    (ta, sa, pa), (tb, sb, pb) = res
    print("case", case, w, h, kind, "K", K, "cs", cs, "weighted", wts is not None, "same trace:", ta == tb,
          "stats host", {k: sa[k] for k in ("n_base_clusters", "n_clusters", "split_evals", "split_px", "lq_rounds")},
          "dev", {k: sb[k] for k in ("n_base_clusters", "n_clusters", "split_evals", "split_px", "lq_rounds")})
    if ta != tb:
        for k in ta:
            if k != "splits" and ta[k] != tb[k]:
                print("   header", k, ta[k], tb[k])
        for i, (a, b) in enumerate(zip(ta["splits"], tb["splits"])):
            if a != b:
                print("   first differing record", i)
                for k in a:
                    print("      %-10s %s %r | %r" % (k, "  " if a[k] == b[k] else "!=", a[k], b[k]))
                break
        print("   records", len(ta["splits"]), len(tb["splits"]))
