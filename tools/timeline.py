"""Timeline of one bench step out of a rocprofv3 --kernel-trace csv: start (us), gap to the previous kernel, duration, name, grid.
usage: timeline.py kernel_trace.csv [step-from-the-end, default 2]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def short(n):
    m = re.search(r"pamd::(\w+)", n)
    return m.group(1) if m else n[:32]


idx = [i for i, r in enumerate(rows) if "k_convert" in r["Kernel_Name"]]
start, end = idx[-back - 1], idx[-back]
t0 = int(rows[start]["Start_Timestamp"])
prev = None
busy = 0.0
for r in rows[start:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    busy += (e - s) / 1e3
    print("%9.1f  gap %7.1f  dur %7.1f  %-24s grid %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, short(r["Kernel_Name"]), r["Grid_Size_X"]))
    prev = e
print("step %.1f us, kernels busy %.1f us" % ((int(rows[end]["Start_Timestamp"]) - t0) / 1e3, busy))
