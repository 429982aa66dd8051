#!/bin/bash
# round 4, call B: merged attention backward -- tests, the lab (split / exchange-tile / default), phase trace of the default kernel
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_00_kernels_gpu.py tests/test_10_model_gpu.py -q -p no:cacheprovider -k "attn or attention or pretext or fixture or extreme or self_spread" > gpurun_out/r4b_tests.log 2>&1; echo "tests exit $?"; tail -n 5 gpurun_out/r4b_tests.log
( for m in ${MODES:-split xch one}; do echo "== VLP_ATTN_BWD=$m"; VLP_ATTN_BWD=$m timeout 300 python tools/attn_lab.py; done
  [ -n "$GRID0" ] && { echo "== one workgroup per item (VLP_ATTN_BWD_GRID=0)"; VLP_ATTN_BWD_GRID=0 timeout 300 python tools/attn_lab.py; } ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4b_attn_lab.log
[ -f vlp_amd/libvlp_hip_trace.so ] && for b in 21 64; do VLP_HIP_LIB=vlp_amd/libvlp_hip_trace.so python tools/attn_bwd_trace.py $b 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r4b_attn_trace.log
