"""SHA-256 of vlp_attn_fwd outputs (context rows + lse) over a fixed set of seeded cases: run under two builds of the library
(VLP_HIP_LIB=...) and diff the output to show that a rewrite of the forward tile leaves every bit where it was.  Also times the B = 64 case.
usage: python tools/attn_fwd_bits.py [--time]"""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K
from vlp_amd import synthetic as S
DEV = torch.device("cuda:0")
A, H = 12, 768


def sha(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def case(B, Nv, max_len_b, s2s_prob, p, skip, packed, seed):
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    batch = S.make_batch(B, max_len_b=max_len_b, len_vis_input=Nv, vocab_size=1000, max_pred=3, s2s_prob=s2s_prob, seed=seed, min_len_b=min(6, max_len_b))
    mask = batch.input_mask.to(DEV)
    L = mask.shape[1]
    Lp = (L + 31) // 32 * 32
    mb = torch.empty(B, L, Lp, dtype=torch.uint8, device=DEV)
    K.mask_pack(mask, mb, B, L, Lp)
    os.environ["VLP_ATTN_SKIP"] = "1" if skip else "0"
    if packed:
        col_any = (mask.sum(1) != 0)                                                # [B, L]: key attended by some query
        lens = (col_any * torch.arange(1, L + 1, device=DEV)[None, :]).amax(1).to(torch.int32)      # kept rows = last attended position + 1
        row_off = torch.zeros(B + 1, dtype=torch.int32, device=DEV); row_off[1:] = torch.cumsum(lens, 0)
        M = int(row_off[-1])
        qkv = (torch.randn(M, 3 * H, device=DEV, generator=g) * 0.5).half()
        ctx = torch.zeros(M, H, device=DEV, dtype=torch.half)
    else:
        row_off = None
        qkv = (torch.randn(B * L, 3 * H, device=DEV, generator=g) * 0.5).half()
        ctx = torch.zeros(B * L, H, device=DEV, dtype=torch.half)
    lse = torch.zeros(B, A, L, device=DEV)
    K.attn_fwd(qkv, mb, ctx, lse, B, L, A, 0.125, dropout_p=p, seed=seed + 1, row_off=row_off)
    torch.cuda.synchronize()
    if packed:      # lse of rows past a sample's kept rows is not written
        keep = torch.arange(L, device=DEV)[None, None, :] < lens[:, None, None].to(DEV)
        lse = torch.where(keep, lse, torch.zeros_like(lse))
    return sha(ctx, lse), (qkv, mb, ctx, lse, B, L)


if __name__ == "__main__":
    print("library:", os.environ.get("VLP_HIP_LIB", "(product)"))
    for (B, Nv, nb, s2s, p, skip, packed, seed) in [(64, 100, 64, 1.0, 0.1, 1, 0, 1), (64, 100, 64, 1.0, 0.1, 0, 0, 1), (64, 100, 64, 0.0, 0.1, 1, 0, 2), (64, 100, 64, 0.75, 0.0, 1, 0, 3),
                                                    (7, 100, 64, 1.0, 0.3, 1, 0, 4), (64, 100, 64, 1.0, 0.1, 1, 1, 5), (32, 100, 64, 0.5, 0.0, 1, 1, 6), (3, 100, 64, 1.0, 0.5, 0, 1, 7),
                                                    (5, 8, 20, 1.0, 0.1, 1, 0, 8), (5, 36, 61, 1.0, 0.1, 1, 0, 9), (5, 36, 61, 0.0, 0.0, 0, 1, 10), (4, 100, 120, 1.0, 0.1, 1, 0, 11),
                                                    (4, 100, 150, 0.5, 0.0, 1, 1, 12), (9, 100, 26, 1.0, 0.2, 1, 0, 13)]:
        d, keepalive = case(B, Nv, nb, s2s, p, skip, packed, seed)
        print("B=%-3d L=%-3d s2s=%.2f p=%.1f skip=%d packed=%d  %s" % (B, Nv + nb + 3, s2s, p, skip, packed, d))
    if "--time" in sys.argv:
        os.environ["VLP_ATTN_SKIP"] = "1"
        g = torch.Generator(device=DEV); g.manual_seed(0)
        B, L = 64, 167
        batch = S.make_batch(B, max_len_b=64, vocab_size=1000, max_pred=3, s2s_prob=1.0, seed=3)
        Lp = (L + 31) // 32 * 32
        mb = torch.empty(B, L, Lp, dtype=torch.uint8, device=DEV)
        K.mask_pack(batch.input_mask.to(DEV), mb, B, L, Lp)
        sets = [((torch.randn(B * L, 3 * H, device=DEV, generator=g) * 0.5).half(), torch.empty(B * L, H, device=DEV, dtype=torch.half), torch.empty(B, A, L, device=DEV)) for _ in range(6)]
        c = [0]

        def fwd(p):
            qkv, ctx, lse = sets[c[0] % 6]; c[0] += 1
            K.attn_fwd(qkv, mb, ctx, lse, B, L, A, 0.125, dropout_p=p, seed=1)
        for p in (0.1, 0.0):
            for _ in range(6): fwd(p)
            torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(60): fwd(p)
            e.record(); torch.cuda.synchronize()
            print("attn_fwd B=64 L=167 dropout %.1f: %.1f us per launch (rotating operand sets)" % (p, s.elapsed_time(e) / 60 * 1e3))
