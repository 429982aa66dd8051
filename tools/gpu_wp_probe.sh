#!/bin/bash
# decomposition of the wave-pipelined loops: product + WP_DBG investigation builds.  usage: gpu_wp_probe.sh variants shapes "masks" [env...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=${1:-27,29,72,73,76,77,78}
S=${2:-768x3072,3072x768}
( python tools/wp_probe.py $V $S
for m in ${3:-4 6 5 7 12 14}; do VLP_HIP_LIB=vlp_amd/libvlp_hip_wpd$m.so python tools/wp_probe.py $V $S; done ) 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/wp_probe.log
