#!/bin/bash
# Investigation builds of the wave-pipelined NT family: vlp_amd/libvlp_hip_wpd<mask>.so = the product objects with gemm_nt_wp.hip
# recompiled under -DWP_DBG=<mask> (bit 0 no MFMAs, 1 no DMA, 2 no epilogue, 3 no fragment reads).  usage: tools/build_wp_dbg.sh 1 2 4 6 ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT && python -m vlp_amd.build >/dev/null
for m in "$@"; do
  (
  T=$(mktemp -d)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $ROOT/include -I $ROOT/vlp_amd/csrc -Wno-unused-result -ffp-contract=fast -DWP_DBG=$m -c $ROOT/vlp_amd/csrc/gemm_nt_wp.hip -o $T/gemm_nt_wp.o 2>/dev/null
  OBJS=$(ls $ROOT/vlp_amd/csrc/build/*.o | grep -v gemm_nt_wp.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/vlp_amd/libvlp_hip_wpd$m.so $OBJS $T/gemm_nt_wp.o
  rm -rf $T; echo built wpd$m
  ) &
done
wait
