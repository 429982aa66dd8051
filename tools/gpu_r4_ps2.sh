#!/bin/bash
# persistent k-stream NT GEMM: tests, lab, in-step A/B of the table entries
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "persistent_stream" > gpurun_out/ps_tests.log 2>&1; echo "tests exit $?"; tail -n 3 gpurun_out/ps_tests.log
timeout 600 python tools/nt_lab.py --rotate=12 --variants=10,29,264 2>&1 | grep -v amdgpu.ids > gpurun_out/ps_lab2.txt; cut -c1-200 gpurun_out/ps_lab2.txt
bash tools/gpu_ab_env.sh "default:" "qkv264:VLP_TUNE_TABLE=tools/ab_tables/qkv264.json" "wide264:VLP_TUNE_TABLE=tools/ab_tables/wide264.json" "default again:" "qkv264 again:VLP_TUNE_TABLE=tools/ab_tables/qkv264.json" | tee gpurun_out/ps_ab.txt
