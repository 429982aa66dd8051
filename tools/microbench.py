"""Per-kernel timing at the BASELINE shapes (B=64, L=167, BERT-base) through the C ABI.
Writes gpurun_out/microbench.json.  Usage on the GPU box:  python tools/microbench.py [--quick]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    quick = "--quick" in sys.argv
    only = [a.split("=")[1] for a in sys.argv if a.startswith("--only=")]
    B, L, H, I, A = 64, 167, 768, 3072, 12
    M = B * L
    res = {}
    g = torch.Generator(device=DEV)
    g.manual_seed(0)

    def r(*s):
        return (torch.randn(*s, device=DEV, generator=g) * 0.5).half()

    # ---- NT GEMMs
    for name, (m, n, k) in {} if '--tn-sweep' in sys.argv else {"qkv": (M, 3 * H, H), "attn_out": (M, H, H), "ffn1": (M, I, H), "ffn2": (M, H, I),
                            "fc7": (B * 100, 2048, 2048), "lm_decoder": (192, 28996, H)}.items():
        x, w = r(m, k), r(n, k)
        ldy = (n + 63) // 64 * 64
        y = torch.empty(m, ldy, device=DEV, dtype=torch.half)
        bias = r(ldy)
        for var in ((19, 27, 3, 11, 13) if '--ph' in sys.argv else (1, 2, 4, 5, 6, 7, 9, 11, 12, 13, 14, 15)):
            us = timeit(lambda: K.gemm_nt(x, w, y, m, n, k, bias=bias, variant=var))
            res["gemm_nt/%s/v%d" % (name, var)] = {"us": us, "tflops": 2.0 * m * n * k / us / 1e6}
        if name == "ffn1":
            z = torch.empty_like(y)
            us = timeit(lambda: K.gemm_nt(x, w, y, m, n, k, bias=bias, preact=z, act=K.ACT_GELU, variant=0))
            res["gemm_nt/ffn1+gelu/v0"] = {"us": us, "tflops": 2.0 * m * n * k / us / 1e6}
        if name == "ffn2":
            rs = r(m, n)
            us = timeit(lambda: K.gemm_nt(x, w, y, m, n, k, bias=bias, residual=rs, dropout_p=0.1, seed=1, variant=0))
            res["gemm_nt/ffn2+drop+res/v0"] = {"us": us, "tflops": 2.0 * m * n * k / us / 1e6}
        a = torch.randn(m, k, device=DEV, dtype=torch.half)
        us = timeit(lambda: torch.matmul(a, w.t()))
        res["torch_matmul/%s" % name] = {"us": us, "tflops": 2.0 * m * n * k / us / 1e6}
    if '--nt-only' in sys.argv:
        json.dump(res, open('gpurun_out/microbench_nt.json', 'w'), indent=1)
        for k, v in res.items():
            print('%-40s %8.1f us  tflops=%.1f' % (k, v['us'], v['tflops']))
        return
    # ---- TN GEMMs (wgrad)
    for name, (m, n, k) in {"w_qkv": (M, 3 * H, H), "w_out": (M, H, H), "w_ffn1": (M, I, H), "w_ffn2": (M, H, I)}.items():
        a, b = r(m, n), r(m, k)
        c = torch.empty(n, k, device=DEV, dtype=torch.half)
        ws = torch.empty(K.gemm_tn_workspace_bytes(m, n, k), device=DEV, dtype=torch.uint8)
        for var in ((2, 10, 26) if '--tn-sweep' in sys.argv else (2, 3, 4, 5)):
            for sp in (range(1, 17) if '--tn-sweep' in sys.argv else ((0, 2, 4, 8) if quick else (0, 1, 2, 4, 8))):
                us = timeit(lambda: K.gemm_tn(a, b, c, m, n, k, workspace=ws, variant=var, splits=sp), iters=10)
                res["gemm_tn/%s/v%d/s%d" % (name, var, sp)] = {"us": us, "tflops": 2.0 * m * n * k / us / 1e6}
        us = timeit(lambda: torch.matmul(a.t(), b))
        res["torch_matmul/%s" % name] = {"us": us, "tflops": 2.0 * m * n * k / us / 1e6}
    # ---- attention
    qkv = r(M, 3 * H)
    Lp = (L + 31) // 32 * 32
    mb = torch.ones(B, L, Lp, device=DEV, dtype=torch.uint8)
    mb[:, :, L:] = 2
    mt = torch.full((B, Lp, Lp), 2, device=DEV, dtype=torch.uint8)
    mt[:, :L, :L] = 1
    ctx, dctx = torch.empty(M, H, device=DEV, dtype=torch.half), r(M, H)
    lse, delta = torch.empty(B, A, L, device=DEV), torch.empty(B, A, L, device=DEV)
    dqkv = torch.empty_like(qkv)
    fl = 4.0 * B * A * L * L * 64
    for p in (0.0, 0.1):
        us = timeit(lambda: K.attn_fwd(qkv, mb, ctx, lse, B, L, A, 0.125, dropout_p=p, seed=1))
        res["attn_fwd/p%.1f" % p] = {"us": us, "tflops": fl / us / 1e6}
        us = timeit(lambda: K.attn_bwd(qkv, mb, mt, ctx, dctx, lse, dqkv, delta, B, L, A, 0.125, dropout_p=p, seed=1))
        res["attn_bwd/p%.1f" % p] = {"us": us, "tflops": 2.5 * fl / us / 1e6}
    # ---- memory-bound kernels
    x = r(M, H)
    gamma, beta = r(H), r(H)
    y = torch.empty_like(x)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    us = timeit(lambda: K.layernorm_fwd(x, gamma, beta, y, M, H, mean, rstd))
    res["layernorm_fwd"] = {"us": us, "GBps": 2.0 * M * H * 2 / us / 1e3}
    dx, dg, db = torch.empty_like(x), torch.empty(H, device=DEV, dtype=torch.half), torch.empty(H, device=DEV, dtype=torch.half)
    ws = torch.empty(K.layernorm_bwd_workspace_bytes(H), device=DEV, dtype=torch.uint8)
    us = timeit(lambda: K.layernorm_bwd(y, x, gamma, mean, rstd, dx, dg, db, M, H, ws))
    res["layernorm_bwd"] = {"us": us, "GBps": 3.0 * M * H * 2 / us / 1e3}
    z = r(M, I)
    out = torch.empty(I, device=DEV, dtype=torch.half)
    ws2 = torch.empty(K.colsum_workspace_bytes(M, I), device=DEV, dtype=torch.uint8)
    us = timeit(lambda: K.colsum(z, out, M, I, workspace=ws2))
    res["colsum_3072"] = {"us": us, "GBps": M * I * 2 / us / 1e3}
    n = 115_000_000 // 8 * 8
    p32, m_, v_ = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    g16, p16 = torch.zeros(n, device=DEV, dtype=torch.half), torch.zeros(n, device=DEV, dtype=torch.half)
    hyper = torch.tensor([1.0, 1e-5, 0.0], device=DEV)
    us = timeit(lambda: K.fused_adam(p32, m_, v_, g16, p16, n, hyper), iters=5)
    res["fused_adam_115M"] = {"us": us, "GBps": n * 28.0 / us / 1e3}
    out2, part = torch.zeros(2, device=DEV), torch.zeros(2048, device=DEV)
    us = timeit(lambda: K.sumsq(g16, n, out2, part), iters=5)
    res["sumsq_115M"] = {"us": us, "GBps": n * 2.0 / us / 1e3}

    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/microbench.json", "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        print("%-36s %10.1f us  %s" % (k, v["us"], "  ".join("%s=%.1f" % (a, b) for a, b in v.items() if a != "us")))


if __name__ == "__main__":
    main()
