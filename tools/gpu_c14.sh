#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_00_kernels_gpu.py tests/test_10_model_gpu.py tests/test_20_fullsize_gpu.py tests/test_30_train_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 2
bash tools/gpu_ab_env.sh "new (defer+shadow side):" "no defer:VLP_LN_DEFER=0" "no shadow side:VLP_SHADOW_SIDE=0" "neither:VLP_LN_DEFER=0;VLP_SHADOW_SIDE=0" "new again:"
