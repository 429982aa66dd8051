#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/blt && mkdir -p /tmp/blt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/blt -o blt --output-format csv -- python $OLDPWD/tools/blaslt_names.py > /tmp/blt/log.txt 2>&1)
f=$(find /tmp/blt -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
cut -c1-1200 "$f" | head -20 | tee gpurun_out/blaslt_names.txt
