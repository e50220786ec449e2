"""More seeds of tests/test_gpu_fuzz.py's sweep (GPU box): every random configuration identical to the oracle, or the difference proven
tie noise by tests/tie_prover.py.  usage: fuzz_proven.py first_seed n_seeds [cases_per_seed=60]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import patolette_amd as p
from oracle import binding as ob
from patolette_amd import _native as native
from tests import test_gpu_fuzz as T

ob.lib()
gpu = native.lib()
first, nseeds = int(sys.argv[1]), int(sys.argv[2])
ncases = int(sys.argv[3]) if len(sys.argv) > 3 else 60
tot_exact = tot = 0
allproven = {}
for seed in range(first, first + nseeds):
    rng = np.random.default_rng(seed)
    exact, proven = 0, {}
    for case in range(ncases):
        w, h, kind, colors, wts, o = T._case(rng)
        try:
            why = T._run_case(p, ob, native, gpu, w, h, colors, wts, o)
        except AssertionError as e:
            print("VIOLATION seed %d case %d %dx%d %s %r weighted=%s: %s" % (seed, case, w, h, kind, o, wts is not None, e), flush=True)
            continue
        if why is None:
            exact += 1
            continue
        if kind in ("noise", "scene"):
            print("GENERIC CONTENT DIFFERS seed %d case %d %s %r: %s" % (seed, case, kind, o, why), flush=True)
        key = (kind, why["first"][0] if why["first"] else "conversion")
        proven[key] = proven.get(key, 0) + 1
        allproven[key] = allproven.get(key, 0) + 1
    tot_exact += exact
    tot += ncases
    print("seed %d: exact %d of %d, proven ties %s" % (seed, exact, ncases, proven), flush=True)
print("total: %d of %d identical, %d proven ties: %s" % (tot_exact, tot, sum(allproven.values()), allproven))
