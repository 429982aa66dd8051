#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/profd
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profd -o dec -- python tools/decode_bench.py > gpurun_out/prof_decode.log 2>gpurun_out/prof_decode.err
python tools/prof_summary.py /tmp/profd/dec_kernel_trace.csv 0.25 > gpurun_out/prof_decode_steady.txt
head -30 gpurun_out/prof_decode_steady.txt | cut -c1-150
