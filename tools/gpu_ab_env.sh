#!/bin/bash
# in-step A/B of environment switches on ONE box.  usage: gpu_ab_env.sh "NAME1:ENV=..|ENV=.." "NAME2:..." ...
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for spec in "$@"; do
  name="${spec%%:*}"; envs="${spec#*:}"
  ( IFS='|'; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
v=d['config'].get('varlen') or {}
print('%-34s %.3f ms/step  %.1f samples/s  NT avg %.2f us (%.1f TF) | varlen %s ms/step' % ('$name', d['ms_per_step'], d['value'], r['avg_launch_us'], r['achieved'], v.get('ms_per_step')))" )
done
