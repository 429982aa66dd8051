"""Fused Adam lab: the 115.9 M-element decay group (3.25 GB of traffic per launch) timed standalone per VLP_ADAM_MODE
(bit 0 = non-temporal state accesses, mode >> 8 = block cap).   python tools/adam_lab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K  # noqa: E402

DEV = torch.device("cuda:0")
N = 115_943_424 // 8 * 8


def main():
    p32 = torch.randn(N, device=DEV) * 0.02
    m, v = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    g = (torch.randn(N, device=DEV) * 0.01).half()
    p16 = p32.half()
    hyper = torch.tensor([1.0, 1e-4, 0.0], device=DEV)
    big = torch.empty(512 * 1024 * 1024 // 2, device=DEV, dtype=torch.float16)      # cache flusher between launches
    bytes_ = N * (4 * 3 * 2 + 2 + 2)
    # yardstick: a plain device copy of the same mix (1:1 read / write) -- what this chip sustains on a streaming read+write pass
    dst = torch.empty_like(p32)
    for name, fn, nbytes in (("torch copy 464 MB fp32 -> fp32", lambda: dst.copy_(p32), N * 8), ("torch p32*1.0001 (in place)", lambda: p32.mul_(1.0001), N * 8)):
        ts = []
        for it in range(6):
            big.fill_(1.0)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        ts = sorted(ts[1:])
        print("%s: median %.1f us -> %.2f TB/s" % (name, ts[len(ts) // 2], nbytes / ts[len(ts) // 2] / 1e6), flush=True)
    for mode in [int(x) for x in os.environ.get('MODES', '').split(',') if x] or (0, 1, (512 << 8) | 1, (768 << 8) | 1, (1024 << 8) | 1, (1280 << 8) | 1, (1536 << 8) | 1, (2048 << 8) | 1):
        os.environ["VLP_ADAM_MODE"] = str(mode)
        ts = []
        for it in range(6):
            big.fill_(1.0)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            K.fused_adam(p32, m, v, g, p16, N, hyper, b1=0.9, b2=0.999, eps=1e-8, decay=0.01, eps_inside_sqrt=False)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        ts = sorted(ts[1:])
        print("mode nt=%d runs=%d cap=%5d: median %.1f us  min %.1f  -> %.2f TB/s" % (mode & 1, 4 if mode & 2 else 2, (mode >> 8) or 4096, ts[len(ts) // 2], ts[0], bytes_ / ts[len(ts) // 2] / 1e6), flush=True)


main()
