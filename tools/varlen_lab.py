"""M-sweep lab for the padding-free (varlen) step: the per-layer kernels of the training step timed at the dense row count
(M = 64 x 167 = 10 688) and at the packed row counts a batch of the SURVEY 8(d) caption-length distribution produces (mean 8 832),
cold operands (rotation over ROT sets), per NT variant.  Tile quantisation decides what packing buys: a launch whose tile count stays
inside the same number of rounds gains nothing.     python tools/varlen_lab.py [--ms=10688,9600,...] [--variants=..]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K  # noqa: E402

DEV = torch.device("cuda:0")
H, I = 768, 3072
ROT = int(os.environ.get("ROT", "6"))


def timeit(fn, iters=24, warm=3):
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
    ms = [10688, 9984, 9472, 8960, 8832, 8448, 7936]
    variants = [77, 73, 29, 27, 264, 10, 26]
    for a in sys.argv:
        if a.startswith("--ms="):
            ms = [int(x) for x in a.split("=")[1].split(",")]
        if a.startswith("--variants="):
            variants = [int(x) for x in a.split("=")[1].split(",")]
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    MX = max(ms)

    def r(*s, scale=0.5):
        return (torch.randn(*s, device=DEV, generator=g) * scale).half()

    cases = [
        ("qkv        N=2304 K=768  bias", 3 * H, H, lambda n: dict(bias=r(n))),
        ("attn_out   N=768  K=768  bias+drop+res", H, H, lambda n: dict(bias=r(n), residual=r(MX, n), dropout_p=0.1, seed=1, rng_stream=2)),
        ("ffn_up     N=3072 K=768  savegrad gelu", I, H, lambda n: dict(bias=r(n), preact=torch.empty(MX, n, device=DEV, dtype=torch.half), act=K.ACT_GELU_SAVE_GRAD)),
        ("ffn_down   N=768  K=3072 bias+drop+res", H, I, lambda n: dict(bias=r(n), residual=r(MX, n), dropout_p=0.1, seed=1, rng_stream=3)),
        ("d_ffn_down N=3072 K=768  plain mul", I, H, lambda n: dict(mul_src=r(MX, n, scale=1.0), mul_mode=K.MUL_PLAIN)),
        ("d_ffn_up   N=768  K=3072 res", H, I, lambda n: dict(residual=r(MX, n))),
        ("d_attn_out N=768  K=768  plain", H, H, lambda n: dict()),
        ("d_qkv      N=768  K=2304 res", H, 3 * H, lambda n: dict(residual=r(MX, n))),
    ]
    res = {}
    for name, n, k, mk in cases:
        sets = [(r(MX, k), r(n, k, scale=0.05), torch.empty(MX, n, device=DEV, dtype=torch.half), mk(n)) for _ in range(ROT)]
        for M in ms:
            row = {}
            for v in variants:
                if v in (29, 27, 264) and n < 1024:
                    continue
                ctr = [0]

                def call(v=v):
                    x, w, y, kw = sets[ctr[0] % ROT]
                    ctr[0] += 1
                    K.gemm_nt(x, w, y, M, n, k, variant=v, **kw)
                try:
                    row[v] = timeit(call)
                except RuntimeError:
                    row[v] = None
            res["%s|%d" % (name, M)] = row
            best = min((u, v) for v, u in row.items() if u)
            print("%-40s M=%5d best v%-3d %6.1f us | %s" % (name, M, best[1], best[0], " ".join("v%d:%.1f" % (v, u) for v, u in row.items() if u)), flush=True)
    # grouped wgrad, LayerNorm
    shapes = [(3 * H, H), (H, H), (I, H), (H, I)]
    sets = [[(r(MX, n), r(MX, k), torch.empty(n, k, device=DEV, dtype=torch.half), torch.empty(n, device=DEV, dtype=torch.half)) for n, k in shapes] for _ in range(ROT)]
    for M in ms:
        i = [0]

        def f():
            s = sets[i[0] % ROT]
            K.gemm_tn_grouped([(a, b, c, M, a.shape[1], b.shape[1], 0, bias) for a, b, c, bias in s])
            i[0] += 1
        us = timeit(f)
        res["tn_grouped|%d" % M] = us
        print("grouped wgrad M=%5d %.1f us" % (M, us), flush=True)
    xs = [r(MX, H) for _ in range(ROT)]
    ys = [torch.empty(MX, H, device=DEV, dtype=torch.half) for _ in range(ROT)]
    gam, bet = r(H), r(H)
    mean, rstd = torch.empty(MX, device=DEV), torch.empty(MX, device=DEV)
    for M in ms:
        i = [0]

        def f():
            K.layernorm_fwd(xs[i[0] % ROT], gam, bet, ys[i[0] % ROT], M, H, mean, rstd)
            i[0] += 1
        us = timeit(f)
        res["ln_fwd|%d" % M] = us
        print("layernorm fwd M=%5d %.1f us" % (M, us), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/varlen_lab.json", "w"), indent=1)


if __name__ == "__main__":
    main()
