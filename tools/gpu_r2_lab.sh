#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python tools/nt_lab.py --tn 2>&1 | tee gpurun_out/nt_lab.txt
bash tools/gpu_pmc_gemm.sh
