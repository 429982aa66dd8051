#!/bin/bash
# round 3: correctness of the wave-pipelined NT family + cold-operand lab against the rings.  usage: tools/gpu_wp_lab.sh [variants] [pytest -k expr]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=${1:-27,29,72,73,74,75,76,77,78,79}
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "${2:-gemm_nt_phased or gemm_nt_asymmetric or gemm_nt_epilogues}" 2>&1 | tail -15 | tee gpurun_out/wp_tests.log
timeout 600 python tools/nt_lab.py --rotate=12 --variants=$V 2>&1 | grep -v amdgpu.ids | tee gpurun_out/wp_lab.log
for m in ${WPD:-}; do VLP_HIP_LIB=vlp_amd/libvlp_hip_wpd$m.so python tools/wp_probe.py 29,72,73,76,77,78 768x3072,3072x768 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/wp_lab.log; done
