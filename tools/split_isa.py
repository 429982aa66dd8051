#!/usr/bin/env python
"""Split a hipcc -S listing into one file per kernel and print MFMA / spill / lane-spill counts (ISA review of the GEMM main loops).
usage: tools/split_isa.py file.s outdir"""
import os
import re
import subprocess
import sys

src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
lines = open(src).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+: ", l)]
for idx, (i, name) in enumerate(starts):
    j = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
    body = lines[i:j]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    tag = re.sub(r"[^A-Za-z0-9]+", "_", dem)[-80:]
    open(os.path.join(out, tag + ".s"), "w").write("\n".join(body))
    meta = [l.strip() for l in body if re.search(r"(sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|next_free_vgpr|accum_offset)", l)]
    print("%-70s lines=%-6d mfma=%-4d writelane=%-4d %s" % (dem[-70:], len(body), sum("v_mfma" in l for l in body), sum("v_writelane" in l for l in body), " ".join(meta[:6])))
