"""Average FETCH_SIZE / WRITE_SIZE per launch of the gemm_nt kernels over the steady-state tail of a bench run.
usage: pmc_summary.py <fetch_counter_csv> <write_counter_csv> <out_json>"""
import csv, json, sys

def tail_avg(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    ids = sorted(int(r["Dispatch_Id"]) for r in rows)
    cut = ids[int(len(ids) * 0.6)]
    vals = [float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) >= cut and "gemm_nt_kernel" in r["Kernel_Name"]]
    return sum(vals) / max(len(vals), 1), len(vals)

f, nf = tail_avg(sys.argv[1], "FETCH_SIZE")
w, nw = tail_avg(sys.argv[2], "WRITE_SIZE")
out = {"kernel": "gemm_nt_kernel (all variants), steady-state tail of bench.py", "launches_sampled": [nf, nw],
       "FETCH_SIZE_KB_raw_avg": f, "WRITE_SIZE_KB_raw_avg": w,
       "note": "gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE as reported",
       "gemm_nt_bytes_per_launch": (2.0 * f + w) * 1024.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
