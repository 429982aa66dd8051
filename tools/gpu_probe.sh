#!/bin/bash
# Runs the per-op GPU parity tests group by group (separate processes) and the microbench; logs under gpurun_out/.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
for grp in gemm_nt gemm_tn colsum attention layernorm "embed or copy2d or vqa" "loss" "adam"; do
  name=$(echo "$grp" | tr ' ' '_')
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "$grp" -p no:cacheprovider > gpurun_out/test_$name.log 2>&1
  echo "== $grp: exit $?" | tee -a gpurun_out/summary.txt
  tail -n 3 gpurun_out/test_$name.log | tee -a gpurun_out/summary.txt
done
timeout 600 python tools/microbench.py --quick > gpurun_out/microbench.log 2>&1
echo "== microbench exit $?" | tee -a gpurun_out/summary.txt
tail -n 60 gpurun_out/microbench.log
