#!/bin/bash
# balanced grouped wgrad (VLP_TN_GROUP_MODE=5): tests, lab (mode 0 / 5 with several tail lengths), in-step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemm_tn" > gpurun_out/sk_tests.log 2>&1; echo "tests exit $?"; tail -n 12 gpurun_out/sk_tests.log
{ VLP_TN_GROUP_MODE=0 timeout 300 python tools/tn_group_lab.py 2>&1 | grep -v amdgpu.ids
for t in ${TAILS:-14 18 21 24 28}; do echo -n "tail $t: "; VLP_TN_SK_TAIL=$t VLP_TN_GROUP_MODE=5 timeout 300 python tools/tn_group_lab.py 2>&1 | grep -v amdgpu.ids; done
VLP_TN_GROUP_MODE=0 timeout 300 python tools/tn_group_lab.py 2>&1 | grep -v amdgpu.ids; } | tee gpurun_out/sk_lab.txt
bash tools/gpu_ab_env.sh "mode0:" "mode5 balanced:VLP_TN_GROUP_MODE=5" "mode0 again:" "mode5 again:VLP_TN_GROUP_MODE=5" | tee gpurun_out/sk_ab.txt
