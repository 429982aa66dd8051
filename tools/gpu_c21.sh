#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_00_kernels_gpu.py tests/test_10_model_gpu.py tests/test_40_decode_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 2
python tools/attn_lab.py 2>/dev/null | grep "B= 64\|B= 42\|B=128"
VLP_HIP_LIB=$PWD/vlp_amd/libvlp_hip_base.so python tools/attn_lab.py 2>/dev/null | grep "B= 64\|B= 42\|B=128"
bash tools/gpu_ab_env.sh "attn prefetch:" "base:VLP_HIP_LIB=$PWD/vlp_amd/libvlp_hip_base.so" "attn prefetch again:" "base again:VLP_HIP_LIB=$PWD/vlp_amd/libvlp_hip_base.so"
