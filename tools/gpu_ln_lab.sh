#!/bin/bash
# LayerNorm kernels: tests + A/B of the half-wave forms (tools/ln_lab.py)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "layernorm or ln_" 2>&1 | tail -5 | tee gpurun_out/ln_tests.log
(VLP_LN_HALFWAVE=0 timeout 300 python tools/ln_lab.py; timeout 300 python tools/ln_lab.py) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ln_lab.log
