#!/bin/bash
# PMC counters of the attention kernels (two passes, <= 8 counters each); summary -> gpurun_out/pmc_attn.txt
mkdir -p gpurun_out; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pa1 /tmp/pa2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pa1 -o p -- python tools/pmc_attn.py > /dev/null 2>gpurun_out/pa1.err; echo "pass1 $?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/pa2 -o p -- python tools/pmc_attn.py > /dev/null 2>gpurun_out/pa2.err; echo "pass2 $?"
python - <<'PY' | tee gpurun_out/pmc_attn.txt
import csv, glob, collections
by = collections.OrderedDict()
for d in ("/tmp/pa1", "/tmp/pa2"):
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0][-28:]
            if "attn" not in name: continue
            by.setdefault((name, int(r["Dispatch_Id"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
# order of dispatches per kernel name: 3 x p=0 then 3 x p=0.1 ; take the last of each triple
names = collections.OrderedDict()
for (name, d), c in by.items():
    names.setdefault(name, []).append((d, c))
for name, lst in names.items():
    lst.sort(key=lambda t: t[0])
    half = len(lst) // 2
    for tag, (d, c) in (("p0.0", lst[half // 2 * 0 + 2 if half >= 3 else 0]), ("p0.1", lst[-1])):
        wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
        print("%-28s %s " % (name, tag) + " ".join("%s=%.3g" % (k.replace("SQ_", ""), v / wc if k != "SQ_WAVE_CYCLES" else v) for k, v in sorted(c.items())))
PY
tail -3 gpurun_out/pa1.err gpurun_out/pa2.err | grep -i "error\|invalid\|not" | head
