#!/bin/bash
# persistent k-stream NT GEMM: tests, cold-operand lab against the rings; MODES = VLP_NT_PS_MODE values to run the lab under
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
: > gpurun_out/ps_lab.txt
for m in ${MODES:-0 2}; do
  echo "== VLP_NT_PS_MODE=$m (tests)" >> gpurun_out/ps_lab.txt
  if [ "$m" != "1" ] && [ "$m" != "3" ]; then
    VLP_NT_PS_MODE=$m timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "persistent_stream" > gpurun_out/ps_tests_$m.log 2>&1; echo "tests mode $m exit $?"; tail -n 3 gpurun_out/ps_tests_$m.log
  fi
  echo "== VLP_NT_PS_MODE=$m" >> gpurun_out/ps_lab.txt
  VLP_NT_PS_MODE=$m timeout 600 python tools/nt_lab.py --rotate=12 --variants=${VARIANTS:-29,264} 2>&1 | grep -v amdgpu.ids >> gpurun_out/ps_lab.txt
done
cut -c1-200 gpurun_out/ps_lab.txt
