#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -p no:cacheprovider -x -k "attention" 2>&1 | tail -n 2
VLP_HIP_LIB=$PWD/vlp_amd/libvlp_hip_trace.so python tools/attn_trace.py 2>/dev/null | grep -v "start ticks\|spread"
python tools/attn_lab.py 2>/dev/null | grep "B= 64"
VLP_HIP_LIB=$PWD/vlp_amd/libvlp_hip_base.so python tools/attn_lab.py 2>/dev/null | grep "B= 64"
