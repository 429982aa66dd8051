#!/bin/bash
# round 4, call D: LayerNorm grid sweeps (rows per wave / software pipeline depth): forward VLP_LN_BLOCKS, backward VLP_LNB_BLOCKS
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "layernorm or ln_" 2>&1 | tail -2
( for fb in 4096 668 446 334 268 256; do for bb in 512 446 384; do [ $bb != 512 ] && [ $fb != 4096 ] && continue; echo "== VLP_LN_BLOCKS=$fb VLP_LNB_BLOCKS=$bb"; VLP_LN_BLOCKS=$fb VLP_LNB_BLOCKS=$bb timeout 300 python tools/ln_lab.py | grep -v HALFWAVE; done; done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4d_ln_lab.log
