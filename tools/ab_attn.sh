#!/bin/bash
# same-box A/B of two builds of libvlp_hip.so: build_ab/lib_base.so vs the in-tree one.  usage (on the GPU box): bash tools/ab_attn.sh [pattern]
PAT=${1:-attn}
for rep in 1 2; do
  for lib in build_ab/lib_base.so vlp_amd/libvlp_hip.so; do
    echo "== $lib"
    VLP_HIP_LIB=$PWD/$lib timeout 300 python tools/microbench.py --quick 2>&1 | grep -E "$PAT"
  done
done
