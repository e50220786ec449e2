#!/usr/bin/env python3
"""Turn gpurun_out/prof/<tag>/ (profiles/collect.sh) into the committed summaries:
  profiles/<tag>_<cfg>_kernel_stats.txt   per-kernel calls / total / avg / min / max from the kernel trace
  profiles/traffic_<cfg>.json             HBM bytes per launch per kernel from the two PMC passes
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for streaming reads
(MI355X_MICROARCH.md, HBM section) and is doubled -- calibrated here on k_convert, whose reads are known
(24 B/px): see "calibration" in the JSON.  WRITE_SIZE is used as is (k_convert writes 24 B/px: matches)."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag, cfg = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof", tag)
outtag = tag[:-len(cfg) - 1] if tag.endswith("_" + cfg) else tag      # collect_kernels.sh: prof/<tag>_<cfg>


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("pamd::", "")
    base = n.split("<")[0]
    if base == "k_hist":
        return "k_hist_gq" if ", true>" in n else "k_hist_lq"
    if base == "k_hist_fix":
        return "k_hist_gq"
    if base == "k_scatter_bin":
        return "k_scatter_cov"
    if base == "k_scatter":
        return "k_scatter_cov" if re.match(r"k_scatter<\w+, true", n) else "k_scatter"      # <W, COV(, INV)>
    if base in ("k_cov_children", "k_cov_nodes"):
        return "k_cov"
    if base in ("k_nn_map_lut", "k_nn_map_mid"):
        return "k_nn_map"
    if base in ("k_km_assign_count", "k_km_assign_lut", "k_km_assign_mid", "k_km_assign_plain", "k_km_assign_sort"):
        return "k_km_assign"
    if base in ("k_km_update", "k_km_update_direct", "k_km_update_lists"):
        return "k_km_update"
    return base


# ---- kernel trace stats
rows = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
with open(os.path.join(root, "profiles", "%s_%s_kernel_stats.txt" % (outtag, cfg)), "w") as out:
    out.write("rocprofv3 --kernel-trace --stats -- python bench.py --config %s --no-cpu-baseline --no-extras --extra-streams 0\n" % cfg)
    out.write("%-86s %7s %12s %10s %10s %10s %6s\n" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        out.write("%-86s %7d %12.1f %10.2f %10.2f %10.2f %6.1f\n" % (k[:86], len(v), sum(v), sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))

# ---- PMC traffic
def load(sub, cname):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname:
                a = agg[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return agg


fe, wr = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
kern = {}
for k in fe:
    n = fe[k][0]
    f_kib, w_kib = fe[k][1] / n, (wr[k][1] / wr[k][0] if k in wr and wr[k][0] else 0.0)
    kern[k] = {"launches": n, "FETCH_SIZE_KiB_per_launch": round(f_kib, 1), "WRITE_SIZE_KiB_per_launch": round(w_kib, 1),
               "hbm_bytes_per_launch": round((2 * f_kib + w_kib) * 1024)}
cal = None
if "k_convert" in kern:
    bench = json.load(open(os.path.join(src, "bench_%s.json" % cfg)))
    npx = bench["config"]["width"] * bench["config"]["height"]
    cal = {"kernel": "k_convert", "known_read_bytes": 24 * npx, "FETCH_SIZE_bytes": kern["k_convert"]["FETCH_SIZE_KiB_per_launch"] * 1024,
           "known_write_bytes": 24 * npx, "WRITE_SIZE_bytes": kern["k_convert"]["WRITE_SIZE_KiB_per_launch"] * 1024}
import datetime
json.dump({"collected": os.environ.get("PAMD_PROFILE_STAMP", datetime.date.today().isoformat()), "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --config %s --steps 2 --warmup 1 --no-cpu-baseline --no-profile --extra-streams 0" % cfg,
           "correction": "hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies 128-B read requests at 64 B)", "calibration": cal, "kernels": kern},
          open(os.path.join(root, "profiles", "traffic_%s.json" % cfg), "w"), indent=1)
for f in ("bench_%s.json" % cfg, "bench_%s_traced.json" % cfg):
    p = os.path.join(src, f)
    if os.path.exists(p):
        open(os.path.join(root, "profiles", "%s_%s" % (outtag, f)), "w").write(open(p).read())
# the kept bench lines were produced BEFORE this traffic file existed (bench.py reads the file of the previous collection): bring
# their roofline.traffic in line with the PMC passes of the same build
traffic = json.load(open(os.path.join(root, "profiles", "traffic_%s.json" % cfg)))
for suffix in ("", "_traced"):
    bp = os.path.join(root, "profiles", "%s_bench_%s%s.json" % (outtag, cfg, suffix))
    if not os.path.exists(bp):
        continue
    lines = open(bp).read().rstrip("\n").split("\n")
    try:
        d = json.loads(lines[-1])
    except ValueError:
        continue
    dom = (d.get("roofline") or {}).get("kernel")
    if dom in traffic.get("kernels", {}):
        d["roofline"]["traffic"] = traffic["kernels"][dom]["hbm_bytes_per_launch"]
        d["roofline"]["traffic_source"] = "profiles/traffic_%s.json (PMC passes of the same build, written after this line was produced)" % cfg
        lines[-1] = json.dumps(d)
        open(bp, "w").write("\n".join(lines) + "\n")
print("written profiles/%s_%s_kernel_stats.txt, profiles/traffic_%s.json" % (outtag, cfg, cfg))
