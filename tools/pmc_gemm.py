"""Launches for the GEMM PMC passes: a few launches of selected (kernel, shape) cases; rocprofv3 attributes counters per dispatch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K  # noqa: E402

DEV = torch.device("cuda:0")
M, H, I = 64 * 167, 768, 3072
g = torch.Generator(device=DEV)
g.manual_seed(0)


def r(*s, scale=0.5):
    return (torch.randn(*s, device=DEV, generator=g) * scale).half()


CASES = os.environ.get("PMC_CASES", "nt:768:3072:11,nt:3072:768:2,nt:3072:768:13,nt:768:768:11,nt:2304:768:10,tn:3072:768:26:3,tn:768:768:26:0").split(",")
for c in CASES:
    f = c.split(":")
    n, k = int(f[1]), int(f[2])
    if f[0] == "nt":
        x, w, y = r(M, k), r(n, k, scale=0.05), torch.empty(M, n, device=DEV, dtype=torch.half)
        for _ in range(3):
            K.gemm_nt(x, w, y, M, n, k, variant=int(f[3]))
    else:
        a, b, cc = r(M, n), r(M, k), torch.empty(n, k, device=DEV, dtype=torch.half)
        ws = torch.empty(K.gemm_tn_workspace_bytes(M, n, k), device=DEV, dtype=torch.uint8)
        for _ in range(3):
            K.gemm_tn(a, b, cc, M, n, k, workspace=ws, variant=int(f[3]), splits=int(f[4]))
torch.cuda.synchronize()
