#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmcA /tmp/pmcB
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmcA -o p -- python tools/pmc_tn.py > /dev/null 2>gpurun_out/pmcA.err; echo "pmcA $?"
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmcA/*counter_collection.csv")
rows = list(csv.DictReader(open(f[0])))
by = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-34:]
    if "gemm" not in name: continue
    key = (r["Dispatch_Id"], name, r.get("Grid_Size", ""))
    by.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
seen = {}
for (d, name, grid), c in by.items():
    k = (name, grid, seen.setdefault((name, grid), 0)); seen[(name, grid)] += 1
    if k[2] % 3 != 2: continue   # third launch of each variant
    wc = c.get("SQ_WAVE_CYCLES", 1)
    print("%-34s #%d conflict/ldsactive=%.2f lds_active/wavecyc=%.3f wait_any=%.2f wait_inst=%.2f active=%.2f mfma_busy/wavecyc=%.3f wait_lds=%.3f" % (
        name, k[2] // 3, c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1), c.get("SQ_LDS_IDX_ACTIVE", 0) / wc,
        c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / wc, c.get("SQ_WAIT_INST_LDS", 0) / wc))
PY
