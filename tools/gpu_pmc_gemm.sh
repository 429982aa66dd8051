#!/bin/bash
# PMC counters of the GEMM kernels (separate passes, <= 8 SQ counters each; no trace domains besides --kernel-trace)
mkdir -p gpurun_out; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pg1 /tmp/pg2 /tmp/pg3 /tmp/pg4
rocprofv3 -L > gpurun_out/rocprof_counters.txt 2>&1
P="python tools/pmc_gemm.py"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/pg1 -o p -- $P > /dev/null 2>gpurun_out/pg1.err; echo "pass1 $?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d /tmp/pg2 -o p -- $P > /dev/null 2>gpurun_out/pg2.err; echo "pass2 $?"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d /tmp/pg3 -o p -- $P > /dev/null 2>gpurun_out/pg3.err; echo "pass3 $?"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/pg4 -o p -- $P > /dev/null 2>gpurun_out/pg4.err; echo "pass4 $?"
python - <<'PY' | tee gpurun_out/pmc_gemm.txt
import csv, glob, collections
by = collections.OrderedDict()
for d in ("/tmp/pg1", "/tmp/pg2", "/tmp/pg3", "/tmp/pg4"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" not in r["Kernel_Name"]: continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gemm_nt_", "")[-44:]
            by.setdefault((name, r.get("Grid_Size", ""), int(r["Dispatch_Id"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
groups = collections.OrderedDict()
for (name, grid, d), c in by.items():
    groups.setdefault((name, grid), []).append((d, c))
for (name, grid), lst in groups.items():
    lst.sort(key=lambda t: t[0])
    c = lst[-1][1]
    wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
    print("%-44s grid=%-8s " % (name, grid) + " ".join("%s=%.4g" % (k.replace("SQ_", ""), (v / wc if k.startswith("SQ_") and k != "SQ_WAVE_CYCLES" else v)) for k, v in sorted(c.items())))
PY
for f in gpurun_out/pg?.err; do grep -i "error\|invalid\|unknown\|not found" $f | head -3; done
