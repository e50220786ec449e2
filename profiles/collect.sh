#!/bin/sh
# Run on the GPU box from the repo root:  sh profiles/collect.sh <tag> <config>
# Produces (under gpurun_out/prof/<tag>/): the bench JSON line, a rocprofv3 --kernel-trace --stats
# summary of the same command, and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) -- never
# combined with other trace domains.  profiles/summarize.py turns them into the committed files.
TAG=${1:-r02}; CFG=${2:-c3}
R=$PWD; OUT=$R/gpurun_out/prof/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
python $R/bench.py --config $CFG > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python $R/bench.py --config $CFG --no-cpu-baseline --no-extras --extra-streams 0 > $OUT/bench_${CFG}_traced.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile --extra-streams 0 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile --extra-streams 0 > /dev/null 2> $OUT/pmc_write.err
cd $R; find $OUT -name "*.csv" | head -20
