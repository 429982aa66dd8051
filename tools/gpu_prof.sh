#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; summary copied to gpurun_out/prof_stats.csv
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.log 2>gpurun_out/prof_bench.err
echo "prof exit $?"
find /tmp/prof -type f | head -20;
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/prof_kernel_stats.csv 2>/dev/null
head -40 gpurun_out/prof_kernel_stats.csv
tail -n 1 gpurun_out/prof_bench.log
