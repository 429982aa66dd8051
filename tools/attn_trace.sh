#!/bin/bash
# investigation: phase timestamps of attn_fwd (build with -DVLP_ATTN_TRACE into a separate library, run, print per-phase cycle statistics)
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d); cd vlp_amd/csrc
for f in *.hip api.cpp; do X=""; [ "${f##*.}" = "cpp" ] && X="-x hip"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVLP_ATTN_TRACE -I ../../include -I . -Wno-unused-result -ffp-contract=fast $X -c $f -o $T/${f%.*}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvlp_hip_trace.so $T/*.o; rm -rf $T; echo built libvlp_hip_trace.so
