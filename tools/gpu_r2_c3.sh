#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_00_kernels_gpu.py tests/test_10_model_gpu.py tests/test_20_fullsize_gpu.py -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_c3.log 2>&1; echo "pytest exit $?"; tail -n 5 gpurun_out/pytest_c3.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_c3.log 2>&1; tail -n 1 gpurun_out/bench_c3.log | cut -c1-400
VLP_WGRAD_SIDE_STREAM=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_c3_noside.log 2>&1; tail -n 1 gpurun_out/bench_c3_noside.log | cut -c1-400
timeout 600 python tools/nt_lab.py --variants=2,10,11,13 --nt-only 2>&1 | tee gpurun_out/nt_lab2.txt | tail -n 12
rm -rf /tmp/prof_c3; VLP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > /dev/null 2>gpurun_out/prof_c3.err; echo "rocprof exit $?"
python tools/prof_summary.py $(find /tmp/prof_c3 -name "*kernel_trace.csv" | head -1) 0.5 > gpurun_out/prof_c3_summary.txt 2>&1; head -n 45 gpurun_out/prof_c3_summary.txt
