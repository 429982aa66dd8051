#!/bin/bash
# round 5: rocprofv3 kernel-trace summaries of the dense and of the padding-free (VLP_VARLEN=1: first leg packed) bench step, side stream off
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd /tmp 2>/dev/null; cd - >/dev/null
for mode in dense varlen; do
  V=0; [ $mode = varlen ] && V=1
  rm -rf /tmp/prof_$mode
  VLP_VARLEN=$V VLP_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-varlen > gpurun_out/prof_$mode.out 2>gpurun_out/prof_$mode.err; echo "rocprof $mode exit $?"
  python tools/prof_summary.py $(find /tmp/prof_$mode -name "*kernel_trace.csv" | head -1) 0.5 > gpurun_out/prof_${mode}_summary.txt 2>&1
  head -n 40 gpurun_out/prof_${mode}_summary.txt
  tail -n 1 gpurun_out/prof_$mode.out | cut -c1-200
done
