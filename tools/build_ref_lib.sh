#!/bin/bash
# Build libvlp_hip of another git revision (same C ABI) next to the current one, for same-box A/B runs through VLP_HIP_LIB.
# usage: tools/build_ref_lib.sh <git-rev> <out.so>
set -e
REV=$1; OUT=$2; ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C $ROOT archive $REV vlp_amd/csrc include | tar -x -C $T
cd $T/vlp_amd/csrc
for f in *.hip api.cpp; do
  X=""; [ "${f##*.}" = "cpp" ] && X="-x hip"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $T/include -I . -Wno-unused-result -ffp-contract=fast $X -c $f -o ${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/$OUT *.o
rm -rf $T
echo built $OUT from $REV
