"""Is the token-step attention bound by its K/V access pattern?  Times vlp_attn_decode (wave-per-(sequence, head) kernel) on the product layout
([B, Lcap, 2H]: a head's rows are 128-byte pieces at a 3 KB stride) and on a head-major-LIKE addressing of the same byte count (row pitch 256 B,
a (sequence, head) item reads one contiguous ~29 KB region), 12 different caches in rotation (cold, as the 12 layers of a step)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vlp_amd import _lib as K
dev = torch.device("cuda:0")
B, T, heads, H, Lcap, Lk = int(os.environ.get("B", 64)), 2, 12, 768, 123, 113
q = torch.randn(B * T, 3 * H, device=dev).half()
mask = torch.ones(B, T, Lk, dtype=torch.long, device=dev)
Lkp = (Lk + 31) // 32 * 32
mb = torch.empty(B, T, Lkp, dtype=torch.uint8, device=dev)
K.mask_pack_rect(mask, mb, B, T, Lk, Lkp)
ctx = torch.empty(B * T, H, device=dev, dtype=torch.float16)
NL = 12
prod = [torch.randn(B, Lcap, 2 * H, device=dev).half() for _ in range(NL)]
hm = [torch.randn(B * heads * Lcap + 64, 128, device=dev).half() for _ in range(NL)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def run(layout, reps=20):
    ts = []
    for _ in range(reps):
        flush.zero_()                      # evict the caches from the Infinity Cache
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(NL):
            if layout == "product":
                kv = prod[i]
                K.attn_decode(q, 3 * H, T, kv, kv[:, :, H:], 2 * H, Lcap, mb, ctx, B, T, Lk, heads, 0.125)
            else:
                kv = hm[i]
                K.attn_decode(q, 3 * H, T, kv, kv[:, 64:], 128, heads * Lcap, mb, ctx, B, T, Lk, heads, 0.125)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / NL)
    ts.sort()
    return ts[len(ts) // 2]


for lay in ("product", "contiguous", "product", "contiguous"):
    print("%-11s %.2f us per launch (B = %d: %.1f MB of K | V per launch)" % (lay, run(lay), B, B * Lk * 2 * H * 2 / 1e6))
