R=$PWD; OUT=$R/gpurun_out/pmcd; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; export PYTHONPATH=$R; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -i -E "SQ[C]?_[A-Z_]*(ICACHE|IFETCH|INST_CACHE)[A-Z_]*" | sort -u | head -20 > $OUT/avail.txt
for c in noise scene; do
  B="python $R/tools/dither_kernels.py 4096"
  DST_CS=2 DK_WEIGHTS=0 DK_CONTENT=$c rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d $OUT/${c}1 -o a --output-format csv -- $B > /dev/null 2> $OUT/${c}1.err
  DST_CS=2 DK_WEIGHTS=0 DK_CONTENT=$c rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQC_ICACHE_MISSES SQC_ICACHE_REQ -d $OUT/${c}2 -o b --output-format csv -- $B > /dev/null 2> $OUT/${c}2.err
done
cd $R
python - <<'PY'
import csv, glob, collections
for c in ("noise", "scene"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("gpurun_out/pmcd/%s[12]/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_dither_lanes<0>" in r["Kernel_Name"] or "k_dither_lanesILi0" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print(c, {k: "%.3g" % (v[1] / max(1, v[0])) for k, v in sorted(agg.items())})
PY
cat gpurun_out/pmcd/avail.txt
