#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
nproc > gpurun_out/host.txt; free -g | head -2 >> gpurun_out/host.txt; lscpu | grep -E "Model name|Socket|Thread|Core" >> gpurun_out/host.txt
timeout 600 python bench.py --steps ${BENCH_STEPS:-20} --warmup 5 > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench exit $?"
tail -n 1 gpurun_out/bench.log; tail -n 3 gpurun_out/bench.err; cat gpurun_out/host.txt
