#!/usr/bin/env python
"""Which hipBLASLt kernels torch picks for the step's NT shapes (run under rocprofv3 --kernel-trace; the kernel NAMES carry the tile configuration)."""
import torch

M, H, I = 64 * 167, 768, 3072
dev = "cuda"
for n, k in ((3 * H, H), (H, H), (I, H), (H, I), (H, 3 * H)):
    x = torch.randn(M, k, device=dev).half()
    w = torch.randn(n, k, device=dev).half()
    for _ in range(5):
        y = torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
