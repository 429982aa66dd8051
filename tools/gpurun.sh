#!/bin/bash
# Build the library locally (hipcc cross-compiles), then run a command on the GPU box.  usage: tools/gpurun.sh <timeout_s> '<cmd>'
set -e
cd "$(dirname "$0")/.."
python -m vlp_amd.build | grep -E "error|linked" || true
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
