#!/bin/bash
# attention kernels: tests + lab (tools/attn_lab.py) with the streaming forward off / on; REF=vlp_amd/libvlp_hip_xxx.so adds a run on another library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_00_kernels_gpu.py tests/test_10_model_gpu.py -x -q -k "attn or attention or fixture or extreme" 2>&1 | tail -4 | tee gpurun_out/attn_tests.log
( [ -n "$REF" ] && { echo "== $REF"; VLP_HIP_LIB=$REF timeout 300 python tools/attn_lab.py; }
  echo "== product, VLP_ATTN_STREAM=0"; VLP_ATTN_STREAM=0 timeout 300 python tools/attn_lab.py
  echo "== product"; timeout 300 python tools/attn_lab.py ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_lab.log
