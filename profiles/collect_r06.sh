#!/bin/sh
# Round 6, final build: the default bench line with its traces (c3), the small image (c2) and configs[3] (c4) as short traced runs,
# and one-step timelines of c2 and c3.  Run on the GPU box from the repo root:  sh profiles/collect_r06.sh
R=$PWD; export TMPDIR=/tmp
sh profiles/collect.sh r06 c3 > /dev/null 2>&1
sh profiles/collect_kernels.sh r06 c2 > /dev/null 2>&1
sh profiles/collect_kernels.sh r06 c4 > /dev/null 2>&1
python bench.py --config c4 > gpurun_out/prof/r06_c4/bench_c4_full.json 2> gpurun_out/prof/r06_c4/bench_c4_full.err
python bench.py --config c2 --no-cpu-baseline > gpurun_out/prof/r06_c2/bench_c2_full.json 2> /dev/null
for cfg in c2 c3; do
  OUT=$R/gpurun_out/prof/r06_tl_$cfg; rm -rf $OUT; mkdir -p $OUT; cd /tmp
  rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $R/bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-profile --extra-streams 0 > /dev/null 2> $OUT/err.txt
  cd $R; f=$(find $OUT -name "*kernel_trace.csv" | head -1); python tools/timeline.py $f 2 > gpurun_out/prof/r06_timeline_$cfg.txt; rm -rf $OUT/trace
done
# keep what travels back small: the summaries are made from the stats / counter csv files
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out/prof
