#!/usr/bin/env python
"""LayerNorm forward / backward lab at the step's shape (M = 64 x 167, H = 768), cold operands (12 rotating sets as the layers of a step).
usage: [VLP_LN_HALFWAVE=0] python tools/ln_lab.py     (run twice to A/B the half-wave kernels)"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K

DEV = "cuda"
M, H, ROT = 64 * 167, 768, 12


def bench(fn, iters=120, warm=12):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g).half()
    sets = []
    for _ in range(ROT):
        sets.append(dict(x=r(M, H), dy=r(M, H), y=torch.empty(M, H, device=DEV, dtype=torch.half), dx=torch.empty(M, H, device=DEV, dtype=torch.half),
                         dxd=torch.empty(M, H, device=DEV, dtype=torch.half), mean=torch.empty(M, device=DEV), rstd=torch.empty(M, device=DEV)))
    gamma, beta = r(H), r(H)
    dgam, dbet = torch.empty(H, device=DEV, dtype=torch.half), torch.empty(H, device=DEV, dtype=torch.half)
    ws = torch.empty(K.layernorm_bwd_workspace_bytes(H), device=DEV, dtype=torch.uint8)
    ctr = [0]

    def fwd(drop):
        s = sets[ctr[0] % ROT]
        ctr[0] += 1
        K.layernorm_fwd(s["x"], gamma, beta, s["y"], M, H, mean=s["mean"], rstd=s["rstd"], eps=1e-12, dropout_p=drop[0], seed=drop[1], rng_stream=drop[2])

    def bwd(dyd, outd, deferred):
        s = sets[ctr[0] % ROT]
        ctr[0] += 1
        K.layernorm_bwd(s["dy"], s["x"], gamma, s["mean"], s["rstd"], s["dx"], dgam, dbet, M, H, ws, dx_drop=s["dxd"] if outd[0] > 0 else None,
                        dy_drop=dyd, out_drop=outd, defer_reduce=deferred)

    for s in sets:
        K.layernorm_fwd(s["x"], gamma, beta, s["y"], M, H, mean=s["mean"], rstd=s["rstd"], eps=1e-12)
    print("VLP_LN_HALFWAVE=%s" % os.environ.get("VLP_LN_HALFWAVE", "1"))
    print("fwd plain            %6.1f us" % bench(lambda: fwd((0.0, 0, 0))))
    print("fwd dropout          %6.1f us" % bench(lambda: fwd((0.1, 1, 2))))
    print("bwd plain  deferred  %6.1f us" % bench(lambda: bwd((0.0, 0, 0), (0.0, 0, 0), True)))
    print("bwd dy-drop deferred %6.1f us" % bench(lambda: bwd((0.1, 1, 2), (0.0, 0, 0), True)))
    print("bwd out-drop deferred%6.1f us" % bench(lambda: bwd((0.0, 0, 0), (0.1, 1, 3), True)))
    print("bwd both   deferred  %6.1f us" % bench(lambda: bwd((0.1, 1, 2), (0.1, 1, 3), True)))
    print("bwd plain  + reduce  %6.1f us" % bench(lambda: bwd((0.0, 0, 0), (0.0, 0, 0), False)))


if __name__ == "__main__":
    main()
