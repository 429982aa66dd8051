#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_00_kernels_gpu.py tests/test_10_model_gpu.py tests/test_40_decode_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 2
bash tools/gpu_ab_env.sh "attn 4 waves:" "attn 8 waves:VLP_ATTN_WAVES=8" "attn 4 waves again:" "attn 8 waves again:VLP_ATTN_WAVES=8"
