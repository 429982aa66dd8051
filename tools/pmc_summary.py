"""Per-kernel FETCH_SIZE / WRITE_SIZE (and optional SQ counters) over the steady-state tail of a rocprofv3 --pmc run of bench.py.
usage: pmc_summary.py <out_json> <counter_csv> [<counter_csv> ...]

gfx950 / ROCm 7.2: FETCH_SIZE under-reports wide coalesced reads (MI355X_MICROARCH.md, HBM section: exactly 1/2 for 16-B/lane
streams).  Instead of assuming the factor it is CALIBRATED in the same run on fused_adam_kernel, whose traffic is known exactly
(reads 4 + 4 + 4 + 2 bytes and writes 4 + 4 + 4 + 2 bytes per parameter element over both flat groups)."""
import collections
import csv
import json
import sys

out_path, paths = sys.argv[1], sys.argv[2:]
rows = []
for p in paths:
    rows += list(csv.DictReader(open(p)))
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
cut = ids[int(len(ids) * 0.5)]


def group(name):
    n = name.split("(")[0]
    if "gemm_nt_wp_kernel" in n or "gemm_nt_ps_kernel" in n:          # the wave-pipelined and persistent families are part of the dominant "vlp_gemm_nt" group
        return "gemm_nt_kernel"
    for key in ("gemm_nt_kernel", "gemm_tn_grouped_kernel", "gemm_tn_glds_kernel", "attn_fwd_kernel", "attn_bwd_full_kernel", "attn_bwd_one_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkv_kernel",
                "fused_adam_kernel", "layernorm_fwd_kernel", "layernorm_bwd_kernel"):
        if key in n:
            return key
    return None


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if int(r["Dispatch_Id"]) < cut:
        continue
    g = group(r["Kernel_Name"])
    if g:
        agg[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"_note": "averages per launch over the steady-state half of the run; *_KB_raw as rocprofv3 reports them"}
for g, c in agg.items():
    res[g] = {k + "_avg": sum(v) / len(v) for k, v in c.items()}
    res[g]["launches"] = max(len(v) for v in c.values())
# calibration on fused_adam (two launches per step: decay group 115.9M elements, no-decay group 0.25M: use the per-launch average of both)
N_ELEMS = None
if "fused_adam_kernel" in res and "FETCH_SIZE_avg" in res["fused_adam_kernel"]:
    a = res["fused_adam_kernel"]
    # elements per launch pair = (decay + nodecay); the average over launches halves it
    known_read_per_elem, known_write_per_elem = 14.0, 14.0
    res["_calibration"] = {"kernel": "fused_adam_kernel", "FETCH_KB_raw_avg": a["FETCH_SIZE_avg"], "WRITE_KB_raw_avg": a.get("WRITE_SIZE_avg"),
                           "read_over_write_raw": a["FETCH_SIZE_avg"] / a["WRITE_SIZE_avg"] if a.get("WRITE_SIZE_avg") else None,
                           "note": "fused Adam reads and writes exactly the same number of bytes (14 per element each), so the true FETCH correction "
                                   "factor for 16-B/lane streams = WRITE_raw / FETCH_raw of this kernel (WRITE_SIZE verified against the known byte count)"}
    if a.get("WRITE_SIZE_avg"):
        factor = a["WRITE_SIZE_avg"] / a["FETCH_SIZE_avg"]
        res["_calibration"]["fetch_factor"] = factor
        for g in res:
            if g.startswith("_") or "FETCH_SIZE_avg" not in res[g]:
                continue
            res[g]["bytes_per_launch"] = (factor * res[g]["FETCH_SIZE_avg"] + res[g].get("WRITE_SIZE_avg", 0.0)) * 1024.0
        if "gemm_nt_kernel" in res:
            res["gemm_nt_bytes_per_launch"] = res["gemm_nt_kernel"].get("bytes_per_launch")
import hashlib      # noqa: E402
import os           # noqa: E402
_h = hashlib.sha256()
_d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vlp_amd", "csrc")
for _f in sorted(os.listdir(_d)):
    if _f.startswith("gemm_nt") or _f == "common.h":
        _h.update(open(os.path.join(_d, _f), "rb").read())
res["kernel_src_sha16"] = _h.hexdigest()[:16]       # bench.py quotes this file only for the same kernel sources
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
