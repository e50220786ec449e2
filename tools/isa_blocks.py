#!/usr/bin/env python3
"""Basic-block instruction census of one kernel in a hipcc -S listing (tools/README.md).
usage: isa_blocks.py file.s kernel-substring"""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(key), l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur, name = [], Counter(), "entry"
order = []
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            order.append((name, cur)); cur, name = Counter(), m.group(1)
        continue
    op = s.split()[0]
    if op.startswith("v_"):
        cls = "valu64" if "f64" in op or "u64" in op or "i64" in op or "b64" in op else "valu"
        if op.startswith("v_pk_"): cls = "vpk"
        if op.startswith("v_cvt"): cls = "vcvt"
    elif op.startswith("ds_"): cls = "lds"
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): cls = "vmem"
    elif op.startswith("s_waitcnt"): cls = "wait"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): cls = "br"
    elif op.startswith("s_"): cls = "salu"
    else: cls = "other"
    cur[cls] += 1
    if cls == "br":
        cur["->" + s.split()[-1]] += 0
order.append((name, cur))
for n, c in order:
    tot = sum(v for k, v in c.items() if not k.startswith("->"))
    tg = " ".join(k for k in c if k.startswith("->"))
    print("%-12s %4d  %s %s" % (n, tot, " ".join("%s=%d" % (k, v) for k, v in sorted(c.items()) if not k.startswith("->")), tg))
