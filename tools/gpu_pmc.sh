#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc1 /tmp/pmc2 /tmp/pmc3
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc1 -o p -- python ${PMC_SCRIPT:-tools/pmc_tn.py} > /dev/null 2>gpurun_out/pmc1.err; echo "pmc1 $?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc2 -o p -- python ${PMC_SCRIPT:-tools/pmc_tn.py} > /dev/null 2>gpurun_out/pmc2.err; echo "pmc2 $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc3 -o p -- python ${PMC_SCRIPT:-tools/pmc_tn.py} > /dev/null 2>gpurun_out/pmc3.err; echo "pmc3 $?"
ls /tmp/pmc1
python - <<'PY'
import csv, glob, collections
for d in ("/tmp/pmc1", "/tmp/pmc2", "/tmp/pmc3"):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f: print(d, "no counter csv", glob.glob(d + "/*")); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"].split("(")[0][-40:]
        if "gemm" not in name and "attn" not in name and "adam" not in name and "layernorm" not in name: continue
        key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "")), r["Counter_Name"])
        v = agg.setdefault(key, [0, 0.0]); v[0] += 1; v[1] += float(r["Counter_Value"])
    for k, v in agg.items():
        print("%-42s grid=%-8s %-14s n=%-3d avg=%.4g" % (k[0], k[1], k[2], v[0], v[1] / v[0]))
PY
