#!/bin/sh
# Run on the GPU box from the repo root:  sh profiles/collect_sq.sh <tag> <config>
# Two separate rocprofv3 --pmc passes of SQ counters (8 slots each; nothing but --kernel-trace next to them) over a short
# bench run.  profiles/summarize_sq.py turns them into profiles/<tag>_<cfg>_sq_counters.txt.
TAG=${1:-r01}; CFG=${2:-c3}
R=$PWD; OUT=$R/gpurun_out/prof/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-profile --extra-streams 0"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d $OUT/sq1 -o a --output-format csv -- $B > /dev/null 2> $OUT/sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_LDS_ATOMIC -d $OUT/sq2 -o b --output-format csv -- $B > /dev/null 2> $OUT/sq2.err
cd $R; find $OUT/sq1 $OUT/sq2 -name "*counter_collection.csv" | head
