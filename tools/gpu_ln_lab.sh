#!/bin/bash
# LayerNorm kernels: tests + lab (tools/ln_lab.py); REF=vlp_amd/libvlp_hip_xxx.so adds a run on another library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "layernorm or ln_" 2>&1 | tail -3 | tee gpurun_out/ln_tests.log
( [ -n "$REF" ] && { echo "== $REF"; VLP_HIP_LIB=$REF timeout 300 python tools/ln_lab.py; }; echo "== product"; timeout 300 python tools/ln_lab.py
  [ -n "$REF" ] && { echo "== $REF again"; VLP_HIP_LIB=$REF timeout 300 python tools/ln_lab.py; } ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ln_lab.log
