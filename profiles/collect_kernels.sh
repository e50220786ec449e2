#!/bin/sh
# Run on the GPU box from the repo root:  sh profiles/collect_kernels.sh <tag> <config>
# For the configurations that are not the default bench line (c4km, c4map ...): kernel-trace stats, the two HBM-traffic
# PMC passes and the two SQ-counter passes of a short run (2 steps, 1 warm-up), each pass on its own.
TAG=${1:-r02}; CFG=${2:-c4km}
R=$PWD; OUT=$R/gpurun_out/prof/${TAG}_$CFG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-extras --extra-streams 0"
$B > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $B > $OUT/bench_${CFG}_traced.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- $B --no-profile > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- $B --no-profile > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d $OUT/sq1 -o a --output-format csv -- $B --no-profile > /dev/null 2> $OUT/sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_LDS_ATOMIC -d $OUT/sq2 -o b --output-format csv -- $B --no-profile > /dev/null 2> $OUT/sq2.err
cd $R; find $OUT -name "*.csv" | head -20
