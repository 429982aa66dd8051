#!/bin/bash
# round 4, call F: rocprofv3 kernel trace of the bench step (side stream off: clean per-kernel times) + one step's timeline with the idle gaps
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
rm -rf /tmp/prof_f; VLP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o p -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events > gpurun_out/r4f_bench_line.json 2>gpurun_out/r4f.err; echo "rocprof exit $?"
T=$(find /tmp/prof_f -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py $T 0.5 > gpurun_out/r4f_summary.txt 2>&1; head -n 30 gpurun_out/r4f_summary.txt
python tools/prof_timeline.py $T > gpurun_out/r4f_timeline.txt 2>&1; tail -n 1 gpurun_out/r4f_timeline.txt
rm -rf /tmp/prof_g; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -o p -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events > gpurun_out/r4f_bench_line_side.json 2>>gpurun_out/r4f.err
T2=$(find /tmp/prof_g -name "*kernel_trace.csv" | head -1)
python tools/prof_timeline.py $T2 > gpurun_out/r4f_timeline_side.txt 2>&1; tail -n 1 gpurun_out/r4f_timeline_side.txt
