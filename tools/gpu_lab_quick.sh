#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "gemm_nt" 2>&1 | tail -n 3
timeout 600 python tools/nt_lab.py --variants=2,9,13,27,59,29,61 --nt-only --rotate=12 2>&1 | tee gpurun_out/nt_lab_rot12c.txt
timeout 600 python tools/nt_lab.py --variants=13,27,59,61 --nt-only --rotate=1 2>&1 | tee gpurun_out/nt_lab_rot1c.txt
