#!/bin/bash
# L2 / vector-memory-path counters of the bench step's kernels (separate --pmc passes) -> gpurun_out/pmc_l2.json
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1 VLP_WGRAD_SIDE_STREAM=0; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-events"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pl$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pl$i -o p -- $B > /dev/null 2>gpurun_out/pmcl$i.err; echo "pass $i ($set): $?"
done
python tools/pmc_summary.py gpurun_out/pmc_l2.json $(find /tmp/pl* -name "*counter_collection.csv") > /dev/null 2>gpurun_out/pmc_l2_summary.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_l2.json'))
for k in ('gemm_nt_kernel','gemm_tn_grouped_kernel','attn_fwd_kernel','fused_adam_kernel','layernorm_bwd_kernel'):
    v=d.get(k)
    if v: print(k, {c:round(x,1) for c,x in v.items()})
PY
tail -3 gpurun_out/pmcl*.err | cut -c1-200 | head -40
