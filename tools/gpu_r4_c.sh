#!/bin/bash
# round 4, call C: full GPU suite after the attention-backward / mask-layout change, then the in-step A/B (two-kernel vs one-kernel backward)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r4c_pytest.log 2>&1; echo "pytest exit $?"; tail -n 12 gpurun_out/r4c_pytest.log
bash tools/gpu_ab_env.sh "split:VLP_ATTN_BWD=split" "one-kernel:VLP_ATTN_BWD=one" "split again:VLP_ATTN_BWD=split" "one-kernel again:VLP_ATTN_BWD=one" 2>&1 | tee gpurun_out/r4c_ab.txt
