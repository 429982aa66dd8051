"""GEMM lab: every per-layer GEMM of the training step at the BASELINE shape (M = 64 x 167 = 10 688), with the epilogue it has in the
step, timed per variant.  Prints a table and writes gpurun_out/nt_lab.json.   python tools/nt_lab.py [--variants=1,2,...] [--tn]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K  # noqa: E402

DEV = torch.device("cuda:0")
M, H, I = 64 * 167, 768, 3072


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
    variants = [1, 2, 3, 9, 10, 11, 12, 13]
    for a in sys.argv:
        if a.startswith("--variants="):
            variants = [int(x) for x in a.split("=")[1].split(",")]
    g = torch.Generator(device=DEV)
    g.manual_seed(0)

    def r(*s, scale=0.5):
        return (torch.randn(*s, device=DEV, generator=g) * scale).half()

    res = {}
    # name: (N, K, epilogue kwargs builder)
    cases = [
        ("qkv        N=2304 K=768  bias", 3 * H, H, lambda n: dict(bias=r(n))),
        ("attn_out   N=768  K=768  bias+drop+res", H, H, lambda n: dict(bias=r(n), residual=r(M, n), dropout_p=0.1, seed=1, rng_stream=2)),
        ("ffn_up     N=3072 K=768  bias+preact+gelu", I, H, lambda n: dict(bias=r(n), preact=torch.empty(M, n, device=DEV, dtype=torch.half), act=K.ACT_GELU)),
        ("ffn_down   N=768  K=3072 bias+drop+res", H, I, lambda n: dict(bias=r(n), residual=r(M, n), dropout_p=0.1, seed=1, rng_stream=3)),
        ("d_ffn_down N=3072 K=768  gelu'mul", I, H, lambda n: dict(mul_src=r(M, n, scale=1.0), mul_mode=K.MUL_GELU_GRAD)),
        ("d_ffn_up   N=768  K=3072 res", H, I, lambda n: dict(residual=r(M, n))),
        ("d_attn_out N=768  K=768  plain", H, H, lambda n: dict()),
        ("d_qkv      N=768  K=2304 res", H, 3 * H, lambda n: dict(residual=r(M, n))),
        ("plain      N=3072 K=768", I, H, lambda n: dict()),
        ("plain      N=768  K=3072", H, I, lambda n: dict()),
        ("plain      N=2304 K=768", 3 * H, H, lambda n: dict()),
    ]
    if "--tn-only" in sys.argv:
        cases = []
    cases.append(("ffn_up     N=3072 K=768  bias+gelu+savegrad", I, H, lambda n: dict(bias=r(n), preact=torch.empty(M, n, device=DEV, dtype=torch.half), act=K.ACT_GELU_SAVE_GRAD)))
    cases.append(("d_ffn_down N=3072 K=768  plain mul", I, H, lambda n: dict(mul_src=r(M, n, scale=1.0), mul_mode=K.MUL_PLAIN)))
    rot = 1
    for a in sys.argv:
        if a.startswith("--rotate="):
            rot = int(a.split("=")[1])       # cycle through `rot` independent operand sets (as the 12 layers of a step do): cold caches
    for name, n, k, mk in cases:
        sets = []
        for _ in range(rot):
            sets.append((r(M, k), r(n, k, scale=0.05), torch.empty(M, n, device=DEV, dtype=torch.half), mk(n)))
        row = {}
        for v in variants:
            if v < 64 and v & 7 == 5 and n < 1024:
                continue
            if 64 <= v < 128 and (v & 7) < 4 and n < 1024:      # wave-pipelined 256x256 tiles: 126 workgroups at N = 768
                continue
            ctr = [0]

            def call(v=v):
                x, w, y, kw = sets[ctr[0] % rot]
                ctr[0] += 1
                K.gemm_nt(x, w, y, M, n, k, variant=v, **kw)
            try:
                us = timeit(call, iters=max(20, 2 * rot))
            except RuntimeError:
                row[v] = None
                continue
            row[v] = us
        ctr = [0]

        def tcall():
            x, w, y, kw = sets[ctr[0] % rot]
            ctr[0] += 1
            torch.matmul(x, w.t())
        us_t = timeit(tcall, iters=max(20, 2 * rot))
        res[name] = {"variants": row, "torch_matmul_plain": us_t, "gflop": 2.0 * M * n * k / 1e9}
        best = min((u, v) for v, u in row.items() if u)
        print("%-42s best v%-2d %6.1f us %6.0f TF | torch plain %6.1f us | %s" % (
            name, best[1], best[0], 2.0 * M * n * k / best[0] / 1e6, us_t, " ".join("v%d:%.1f" % (v, u) for v, u in row.items() if u)))
    if "--tn" in sys.argv:
        for name, n, k in (("w_qkv", 3 * H, H), ("w_out", H, H), ("w_ffn1", I, H), ("w_ffn2", H, I)):
            a, b = r(M, n), r(M, k)
            c = torch.empty(n, k, device=DEV, dtype=torch.half)
            bias = torch.empty(n, device=DEV, dtype=torch.half)
            ws = torch.empty(K.gemm_tn_workspace_bytes(M, n, k), device=DEV, dtype=torch.uint8)
            row = {}
            for var in (2, 26, 3, 27, 4, 28):
                for sp in ((0, 2, 3, 4, 6, 7, 8, 14) if var in (2, 26) else (0, 2, 3, 4, 5, 6, 7, 8)):
                    row["%d/%d" % (var, sp)] = timeit(lambda: K.gemm_tn(a, b, c, M, n, k, workspace=ws, variant=var, splits=sp, bias_out=bias), iters=10)
            us_t = timeit(lambda: torch.matmul(a.t(), b))
            best = min((u, v) for v, u in row.items())
            res["tn/" + name] = {"variants": row, "torch_matmul_plain": us_t, "gflop": 2.0 * M * n * k / 1e9}
            print("tn %-8s best %-5s %6.1f us %6.0f TF | torch %6.1f us | %s" % (name, best[1], best[0], 2.0 * M * n * k / best[0] / 1e6, us_t,
                                                                               " ".join("%s:%.1f" % kv for kv in row.items())))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/nt_lab.json", "w"), indent=1)


if __name__ == "__main__":
    main()
