import json, sys
d = json.load(open(sys.argv[1]))
names = sys.argv[2:] or list(d["kernels"])[:6]
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["achieved"], " ".join("%s=%.3fms/%.0fGB/s" % (k, d["kernels"][k]["ms_per_step"], d["kernels"][k]["GBps"]) for k in names if k in d["kernels"]))
