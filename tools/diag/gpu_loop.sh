mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_gpu_split_loop.py -x -q -s > gpurun_out/r06b/loop.log 2>&1; echo loop rc $?; tail -12 gpurun_out/r06b/loop.log
sh tools/diag/ab_loop.sh
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/r06b/tl; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python $R/bench.py --config c2 --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-profile --extra-streams 0 > /dev/null 2> $OUT/err.txt
cd $R; f=$(find $OUT -name "*kernel_trace.csv" | head -1); python tools/timeline.py $f 2 > gpurun_out/r06b/c2_timeline_dev.txt; tail -2 gpurun_out/r06b/c2_timeline_dev.txt
