#!/bin/bash
# sustained throughput: the bench step for ~30 s (3 000 steps) between two default-length runs on the same box, with clocks / power / temperature around it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
smi() { rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|memory)" | sed 's/^/    /'; }
{
echo "# $(date -u +%FT%TZ) before"; smi
for K in 20 3000 20; do
  timeout 600 python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-parity --no-kernel-events 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['config'].get('varlen') or {}
print('steps %5d  dense %.3f ms/step  %.1f samples/s  loss %.4f  scale %g  skipped %d | padding-free %s ms/step  %s samples/s  loss %s' % (d['steps'], d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['loss_scale'], d['config']['skipped_steps'], v.get('ms_per_step'), v.get('value'), v.get('final_loss')))"
  echo "# after $K steps"; smi
done
} 2>&1 | tee gpurun_out/soak.txt
