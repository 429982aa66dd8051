#!/bin/bash
# A/B of investigation builds against the product library on the N = 3072 / 2304 / 768 shapes: tools/gpu_wp_probe2.sh "<masks>" "<variants>" "<shapes>"
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=${2:-73,77}
S=${3:-3072x768,2304x768,768x3072,768x768}
(timeout 300 python tools/wp_probe.py $V $S
for m in $1; do VLP_HIP_LIB=vlp_amd/libvlp_hip_wpd$m.so timeout 300 python tools/wp_probe.py $V $S; done) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/wp_probe2.log
