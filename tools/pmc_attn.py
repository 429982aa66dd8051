import sys, torch
sys.path.insert(0, ".")
from vlp_amd import _lib as K
DEV = torch.device("cuda:0")
B, L, A, H = 64, 167, 12, 768
g = torch.Generator(device=DEV); g.manual_seed(0)
qkv = (torch.randn(B * L, 3 * H, device=DEV, generator=g) * 0.5).half()
mask = torch.ones(B, L, L, dtype=torch.long, device=DEV)
Lp = (L + 31) // 32 * 32
mb = torch.empty(B, L, Lp, dtype=torch.uint8, device=DEV); mt = torch.empty(B, Lp, Lp, dtype=torch.uint8, device=DEV)
K.mask_pack(mask, mb, B, L, Lp, out_t=mt)
ctx = torch.empty(B * L, H, device=DEV, dtype=torch.half); lse = torch.empty(B, A, L, device=DEV)
dctx = (torch.randn(B * L, H, device=DEV, generator=g) * 0.1).half()
dqkv = torch.empty_like(qkv); delta = torch.empty(B, A, L, device=DEV)
for p in (0.0, 0.1):
    for _ in range(3):
        K.attn_fwd(qkv, mb, ctx, lse, B, L, A, 0.125, dropout_p=p, seed=1)
        K.attn_bwd(qkv, mb, mt, ctx, dctx, lse, dqkv, delta, B, L, A, 0.125, dropout_p=p, seed=1)
torch.cuda.synchronize()
