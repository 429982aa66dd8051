#!/bin/bash
# full GPU check: all -m gpu tests, smoke, short bench
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -n 15 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -n 3 gpurun_out/smoke.log
timeout 900 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 > gpurun_out/bench.log 2>gpurun_out/bench.err; echo "bench exit $?"
tail -n 2 gpurun_out/bench.log; tail -n 5 gpurun_out/bench.err
