#!/bin/bash
# bench (default) + bench without the wgrad side stream + rocprofv3 kernel-trace summary of the same command (side stream off: clean per-kernel times)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
TAG=${1:-x}
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1; tail -n 1 gpurun_out/bench_$TAG.log | cut -c1-180; tail -n 1 gpurun_out/bench_$TAG.log | grep -o '"roofline.*' | cut -c1-400
VLP_WGRAD_SIDE_STREAM=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_${TAG}_noside.log 2>&1; tail -n 1 gpurun_out/bench_${TAG}_noside.log | cut -c1-180
rm -rf /tmp/prof_$TAG; VLP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > /dev/null 2>gpurun_out/prof_$TAG.err; echo "rocprof exit $?"
python tools/prof_summary.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) 0.5 > gpurun_out/prof_${TAG}_summary.txt 2>&1; head -n 26 gpurun_out/prof_${TAG}_summary.txt
