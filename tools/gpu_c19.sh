#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_10_model_gpu.py tests/test_30_train_gpu.py tests/test_60_data_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 3
bash tools/gpu_ab_env.sh "adam pipelined:" "adam plain:VLP_ADAM_PIPELINE=0" "adam pipelined again:" "adam plain again:VLP_ADAM_PIPELINE=0"
