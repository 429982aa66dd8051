#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the kernels of the bench step -> gpurun_out/pmc_traffic.json, pmc_sq.json
mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONUNBUFFERED=1 VLP_WGRAD_SIDE_STREAM=0; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pf /tmp/pw /tmp/ps1 /tmp/ps2
B="python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-parity --no-kernel-events --no-varlen"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o p -- $B > /dev/null 2>gpurun_out/pmcf.err; echo "fetch pass $?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o p -- $B > /dev/null 2>gpurun_out/pmcw.err; echo "write pass $?"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d /tmp/ps1 -o p -- $B > /dev/null 2>gpurun_out/pmcs1.err; echo "sq pass 1 $?"
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/ps2 -o p -- $B > /dev/null 2>gpurun_out/pmcs2.err; echo "sq pass 2 $?"
python tools/pmc_summary.py gpurun_out/pmc_traffic.json $(find /tmp/pf /tmp/pw -name "*counter_collection.csv") > /dev/null; python -c "
import json; d=json.load(open('gpurun_out/pmc_traffic.json')); print(json.dumps({k:v for k,v in d.items() if k in ('_calibration','gemm_nt_kernel','gemm_tn_grouped_kernel','gemm_nt_bytes_per_launch')}, indent=1))"
python tools/pmc_summary.py gpurun_out/pmc_sq.json $(find /tmp/ps1 /tmp/ps2 -name "*counter_collection.csv") > /dev/null; python -c "
import json; d=json.load(open('gpurun_out/pmc_sq.json'))
for k,v in d.items():
    if k.startswith('_') or not isinstance(v, dict): continue
    wc=v.get('SQ_WAVE_CYCLES_avg',0) or 1
    print('%-26s' % k, ' '.join('%s=%.4g' % (c.replace('SQ_','').replace('_avg',''), (x/wc if c.startswith('SQ_') and 'WAVE_CYCLES' not in c else x)) for c,x in sorted(v.items()) if c!='launches'))"
grep -il "error\|invalid" gpurun_out/pmc*.err | head
