#!/bin/bash
# round 2, call 1: full GPU suite (no -x: see everything), smoke, deterministic bench, autotune runs that dump the shape table
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/box.txt; nproc >> gpurun_out/box.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/pytest_gpu_full.log 2>&1; echo "pytest exit $?"
tail -n 40 gpurun_out/pytest_gpu_full.log
timeout 300 python __graft_entry__.py > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -n 2 gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_det.log 2>gpurun_out/bench_det.err; echo "bench(det) exit $?"; tail -n 1 gpurun_out/bench_det.log
rm -f gpurun_out/tuned.json
VLP_AUTOTUNE=1 VLP_TUNE_DUMP=gpurun_out/tuned.json VLP_DEBUG_TUNE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_tune.log 2>gpurun_out/bench_tune.err; echo "bench(tune) exit $?"; tail -n 1 gpurun_out/bench_tune.log
VLP_AUTOTUNE=1 VLP_TUNE_DUMP=gpurun_out/tuned.json timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --tasks vqa2 --s2s_prob 0 > gpurun_out/bench_vqa_tune.log 2>gpurun_out/bench_vqa_tune.err; echo "bench(vqa tune) exit $?"; tail -n 1 gpurun_out/bench_vqa_tune.log
VLP_AUTOTUNE=1 VLP_TUNE_DUMP=gpurun_out/tuned.json timeout 600 python tools/decode_bench.py > gpurun_out/decode_tune.log 2>&1; echo "decode(tune) exit $?"; tail -n 3 gpurun_out/decode_tune.log
timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --s2s_prob 0.75 > gpurun_out/bench_cc.log 2>gpurun_out/bench_cc.err; echo "bench(cc) exit $?"; tail -n 1 gpurun_out/bench_cc.log
