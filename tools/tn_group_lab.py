"""Grouped wgrad lab: the four weight gradients of one BertLayer at the BASELINE shape (M = 10 688) in one vlp_gemm_tn_grouped launch, operand sets
rotated so every launch reads cold data (as in the step).  VLP_TN_GROUP_MODE selects tile shape / ring depth (read once per process).
   for m in 0 1 2 3; do VLP_TN_GROUP_MODE=$m python tools/tn_group_lab.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K  # noqa: E402

DEV = torch.device("cuda:0")
M, H, I = 64 * 167, 768, 3072
ROT = int(os.environ.get("ROT", "6"))


def main():
    g = torch.Generator(device=DEV)
    g.manual_seed(0)

    def r(*s):
        return (torch.randn(*s, device=DEV, generator=g) * 0.25).half()

    shapes = [(3 * H, H), (H, H), (I, H), (H, I)]           # (N of dY, K of X): w_qkv, w_out, w_ffn1, w_ffn2
    sets = []
    for _ in range(ROT):
        sets.append([(r(M, n), r(M, k), torch.empty(n, k, device=DEV, dtype=torch.half), torch.empty(n, device=DEV, dtype=torch.half)) for n, k in shapes])
    i = [0]

    tiles = sum((n // 128) * (k // 128) for n, k in shapes)
    wsk = torch.empty(K.gemm_tn_grouped_workspace_bytes(tiles), device=DEV, dtype=torch.uint8) if K.lab_build() else None      # stream-K form (mode 5, investigation library)

    def f():
        s = sets[i[0] % ROT]
        K.gemm_tn_grouped([(a, b, c, M, a.shape[1], b.shape[1], 0, bias) for a, b, c, bias in s], workspace=wsk)
        i[0] += 1

    for _ in range(3):
        f()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    n = 24
    for _ in range(n):
        f()
    en.record()
    torch.cuda.synchronize()
    us = st.elapsed_time(en) / n * 1e3
    gf = sum(2.0 * M * a * b for a, b in shapes) / 1e9
    # correctness of the first set against torch (fp32 accumulate)
    a, b, c, bias = sets[(i[0] - 1) % ROT][2]
    ref = (a.float().t() @ b.float())
    err = float((c.float() - ref).abs().max() / ref.abs().max())
    print("mode %s  %.1f us per layer  %.0f TFLOP/s  (w_ffn1 rel err %.2e)" % (os.environ.get("VLP_TN_GROUP_MODE", "0"), us, gf / us * 1e3, err))


main()
