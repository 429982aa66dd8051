#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_00_kernels_gpu.py tests/test_10_model_gpu.py tests/test_20_fullsize_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 4
bash tools/gpu_ab_env.sh "grouped+side:" "split+side:VLP_GROUPED_WGRAD=0" "grouped noside:VLP_WGRAD_SIDE_STREAM=0" "split noside:VLP_GROUPED_WGRAD=0;VLP_WGRAD_SIDE_STREAM=0" "grouped+side again:"
