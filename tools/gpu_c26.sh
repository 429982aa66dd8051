#!/bin/bash
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_10_model_gpu.py tests/test_20_fullsize_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 2
bash tools/gpu_ab_env.sh "now:" "now again:"
