#!/usr/bin/env python
"""VGPR / AGPR / scratch (spills) / occupancy / LDS per kernel of the HIP sources (hipcc -Rpass-analysis=kernel-resource-usage).
Run after ANY kernel edit:  python tools/check_resources.py [file.hip ...]   (exit code 1 if a kernel uses scratch)"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vlp_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
bad = 0
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(CSRC, "..", "..", "include"), "-I", CSRC,
                        "-ffp-contract=fast", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, f), "-o", "/tmp/_res.o"],
                       capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        cur["Spill" if k == "VGPRs Spill" else k.split(" ")[0]] = v
        if k.startswith("LDS"):
            flag = "  <-- SCRATCH" if cur.get("ScratchSize", "0") != "0" else ""
            bad += bool(flag)
            print("%-20s %-70s vgpr=%-4s agpr=%-4s spill=%-4s scratch=%-4s occ=%-2s lds=%s%s" % (f, cur["name"][-70:], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("Spill"), cur.get("ScratchSize"),
                                                                                  cur.get("Occupancy"), cur.get("LDS"), flag))
sys.exit(1 if bad else 0)
