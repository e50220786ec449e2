#!/usr/bin/env python3
"""gpurun_out/prof/<tag>/sq{1,2} (profiles/collect_sq.sh) -> profiles/<tag>_<cfg>_sq_counters.txt: per kernel, averaged per
launch, where the wavefronts' cycles go (SQ_WAIT_ANY = parked on s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall,
SQ_ACTIVE_INST_* = issuing; the SQ counters tick in quad-cycles) and the instruction mix."""
import collections
import csv
import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
tag, cfg = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof", tag)
outtag = tag[:-len(cfg) - 1] if tag.endswith("_" + cfg) else tag      # collect_kernels.sh: prof/<tag>_<cfg>


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("pamd::", "").replace("(anonymous namespace)::", "")
    base = n.split("<")[0]
    if base == "k_hist":
        return "k_hist_gq" if ", true>" in n else "k_hist_lq"
    if base == "k_hist_fix":
        return "k_hist_gq"
    if base == "k_scatter_bin":
        return "k_scatter_cov"
    if base == "k_scatter":
        return "k_scatter_cov" if re.match(r"k_scatter<\w+, true", n) else "k_scatter"      # <W, COV(, INV)>
    return base


agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for sub in ("sq1", "sq2"):
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
rows = []
for k, c in agg.items():
    g = lambda n: (c[n][1] / c[n][0]) if n in c and c[n][0] else 0.0   # noqa: E731
    tot = g("SQ_WAIT_ANY") + g("SQ_WAIT_INST_ANY") + g("SQ_ACTIVE_INST_ANY")
    if tot <= 0 or k.startswith("__amd"):
        continue
    rows.append((g("SQ_BUSY_CYCLES"), k, g("SQ_WAVES"), g("SQ_WAIT_ANY") / tot, g("SQ_WAIT_INST_ANY") / tot, g("SQ_ACTIVE_INST_ANY") / tot,
                 g("SQ_ACTIVE_INST_VALU") / tot, g("SQ_ACTIVE_INST_VMEM") / tot, g("SQ_ACTIVE_INST_LDS") / tot,
                 g("SQ_INSTS_VALU"), g("SQ_INSTS_VMEM"), g("SQ_INSTS_LDS"), g("SQ_INSTS_LDS_ATOMIC"),
                 (g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")) if g("SQ_LDS_IDX_ACTIVE") else 0.0, g("SQ_VMEM_TA_ADDR_FIFO_FULL")))
rows.sort(reverse=True)
out = os.path.join(root, "profiles", "%s_%s_sq_counters.txt" % (outtag, cfg))
with open(out, "w") as f:
    f.write("rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes), bench.py --config %s --steps 2 --warmup 1; per launch averages\n" % cfg)
    f.write("shares of wavefront time: parked = SQ_WAIT_ANY, stall = SQ_WAIT_INST_ANY, issue = SQ_ACTIVE_INST_ANY (valu / vmem / lds are parts of it)\n")
    f.write("%-16s %9s %7s %6s %6s %6s %6s %6s %11s %10s %10s %10s %8s %10s\n" % (
        "kernel", "waves", "parked", "stall", "issue", "valu", "vmem", "lds", "valu_insts", "vmem_insts", "lds_insts", "lds_atomic", "lds_conf", "ta_fifo_full"))
    for r in rows:
        f.write("%-16s %9.0f %7.2f %6.2f %6.2f %6.2f %6.2f %6.2f %11.3g %10.3g %10.3g %10.3g %8.2f %10.3g\n" % (r[1], *r[2:]))
print(open(out).read())
