"""Attention kernels vs batch size (workgroups = B x 12): how much is round quantisation / latency, how much throughput?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K
from vlp_amd import synthetic as S
DEV = torch.device("cuda:0")
L, A, H = 167, 12, 768
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
g = torch.Generator(device=DEV); g.manual_seed(0)
for B in (21, 32, 42, 43, 64, 85, 86, 128):
    sets = []
    for r in range(6):          # rotate operand sets: cold caches as in a step
        qkv = (torch.randn(B * L, 3 * H, device=DEV, generator=g) * 0.5).half()
        dctx = (torch.randn(B * L, H, device=DEV, generator=g) * 0.1).half()
        sets.append((qkv, dctx, torch.empty(B * L, H, device=DEV, dtype=torch.half), torch.empty(B, A, L, device=DEV), torch.empty_like(qkv), torch.empty(B, A, L, device=DEV)))
    batch = S.make_batch(B, max_len_b=64, vocab_size=1000, max_pred=3, s2s_prob=1.0, seed=3)
    mask = batch.input_mask.to(DEV)
    Lp = (L + 31) // 32 * 32
    mb = torch.empty(B, L, Lp, dtype=torch.uint8, device=DEV); mt = torch.empty(B, Lp, Lp, dtype=torch.uint8, device=DEV)
    K.mask_pack(mask, mb, B, L, Lp, out_t=mt)
    c = [0]
    def fwd():
        qkv, dctx, ctx, lse, dqkv, delta = sets[c[0] % 6]; c[0] += 1
        K.attn_fwd(qkv, mb, ctx, lse, B, L, A, 0.125, dropout_p=0.1, seed=1)
    def bwd():
        qkv, dctx, ctx, lse, dqkv, delta = sets[c[0] % 6]; c[0] += 1
        K.attn_bwd(qkv, mb, mt, ctx, dctx, lse, dqkv, delta, B, L, A, 0.125, dropout_p=0.1, seed=1)
    for s_ in sets:
        K.attn_fwd(s_[0], mb, s_[2], s_[3], B, L, A, 0.125, dropout_p=0.1, seed=1)
    tf, tb = timeit(fwd), timeit(bwd)
    print("B=%3d  WGs=%4d  fwd %6.1f us (%.3f us/WG-slot-round)  bwd %6.1f us   per-sample fwd %.3f bwd %.3f" % (B, B * A, tf, tf / max(1, -(-B * A // 512)), tb, tf / B, tb / B))
