#!/bin/bash
# Build libvlp_hip from the WORKING TREE with extra compiler flags (investigation builds, e.g. -DVLP_NT_DEBUG) next to the product library.
# usage: tools/build_variant_lib.sh <out.so> <flags...>     then  VLP_HIP_LIB=<out.so> python tools/nt_lab.py ...
set -e
OUT=$1; shift; ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cd $ROOT/vlp_amd/csrc
for f in *.hip api.cpp; do
  X=""; [ "${f##*.}" = "cpp" ] && X="-x hip"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $ROOT/include -I . -Wno-unused-result -ffp-contract=fast "$@" $X -c $f -o $T/${f%.*}.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/$OUT $T/*.o
rm -rf $T
echo built $OUT with "$@"
