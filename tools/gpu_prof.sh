#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; summaries copied to gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps ${BENCH_STEPS:-30} --warmup 5 --no-cpu-baseline > gpurun_out/prof_bench.log 2>gpurun_out/prof_bench.err
echo "prof exit $?"
cp /tmp/prof/bench_kernel_stats.csv gpurun_out/prof_kernel_stats.csv
python tools/prof_summary.py /tmp/prof/bench_kernel_trace.csv 0.4 > gpurun_out/prof_steady.txt
python tools/prof_timeline.py /tmp/prof/bench_kernel_trace.csv > gpurun_out/prof_timeline.txt
cat gpurun_out/prof_steady.txt
tail -n 1 gpurun_out/prof_bench.log | cut -c1-400
