#!/bin/bash
# L2 <-> CU request counters of the kernels of the bench step (separate passes; VERDICT r3 #8: the 8.7x figure was round 2's) -> gpurun_out/pmc_l2.json
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1 VLP_WGRAD_SIDE_STREAM=0; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pl1 /tmp/pl2
B="python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-events"
timeout 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pl1 -o p -- $B > /dev/null 2>gpurun_out/pl1.err; echo "pass 1 $?"
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d /tmp/pl2 -o p -- $B > /dev/null 2>gpurun_out/pl2.err; echo "pass 2 $?"
python tools/pmc_summary.py gpurun_out/pmc_l2.json $(find /tmp/pl1 /tmp/pl2 -name "*counter_collection.csv") > /dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_l2.json"))
for k, v in d.items():
    if not isinstance(v, dict) or "TCP_TCC_READ_REQ_sum_avg" not in v: continue
    rd = v["TCP_TCC_READ_REQ_sum_avg"]; hit = v.get("TCC_HIT_sum_avg", 0); miss = v.get("TCC_MISS_sum_avg", 0)
    print("%-26s launches %4d  L2 read requests %.3g (x 64 B = %.0f MB, x 128 B = %.0f MB)  write requests %.3g  TCC hit %.3g miss %.3g (hit rate %.2f)  EA rd %.3g wr %.3g" % (
        k, v["launches"], rd, rd * 64 / 1e6, rd * 128 / 1e6, v.get("TCP_TCC_WRITE_REQ_sum_avg", 0), hit, miss, hit / max(hit + miss, 1), v.get("TCC_EA0_RDREQ_sum_avg", 0), v.get("TCC_EA0_WRREQ_sum_avg", 0)))
PY
grep -il "error\|invalid" gpurun_out/pl?.err | head
