#!/bin/bash
# in-step A/B of NT variant choices on ONE box: ms/step and the event-sampled NT average per configuration
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 VLP_WGRAD_SIDE_STREAM=0
run() {
  name=$1; shift
  VLP_NT_OVERRIDE="$1" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s %.3f ms/step  NT avg %.2f us  (%.1f TF)' % ('$name', d['ms_per_step'], r['avg_launch_us'], r['achieved']))"
}
W="10688,3072,768"; Q="10688,2304,768"; A="10688,768,768"; B="10688,768,2304"; C="10688,768,3072"
run "table(27/29)" ""
run "old(11,11,9,10,2)" "$A=11;$C=11;$B=9;$Q=10;$W=2"
run "n768=27 wide=old" "$Q=10;$W=2"
run "n768=27 W=13 Q=10" "$Q=10;$W=13"
run "n768=27 W=29 Q=10" "$Q=10"
run "n768=27 W=2 Q=29" "$W=2"
run "n768=19 wide=old" "$A=19;$B=19;$C=19;$Q=10;$W=2"
run "n768=9 wide=old" "$A=9;$B=9;$C=9;$Q=10;$W=2"
run "table(27/29) again" ""
