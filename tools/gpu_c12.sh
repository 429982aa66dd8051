#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "gemm" 2>&1 | tail -n 2
bash tools/gpu_ab_env.sh "base:VLP_HIP_LIB=$PWD/vlp_amd/libvlp_hip_base.so" "epi:VLP_HIP_LIB=$PWD/vlp_amd/libvlp_hip_epi.so" "epi+tnring:" "base again:VLP_HIP_LIB=$PWD/vlp_amd/libvlp_hip_base.so" "epi again:VLP_HIP_LIB=$PWD/vlp_amd/libvlp_hip_epi.so" "epi+tnring again:"
