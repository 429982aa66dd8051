"""Phase timestamps of attn_bwd_full_kernel from the trace build (VLP_HIP_LIB=vlp_amd/libvlp_hip_trace.so, tools/attn_trace.sh)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K
from vlp_amd import synthetic as S
DEV = torch.device("cuda:0"); L, A, H = 167, 12, 768
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator(device=DEV); g.manual_seed(0)
batch = S.make_batch(B, max_len_b=64, vocab_size=1000, max_pred=3, s2s_prob=1.0, seed=3)
Lp = 192
mb = torch.empty(B, L, Lp, dtype=torch.uint8, device=DEV); mt = torch.empty(B, Lp, Lp, dtype=torch.uint8, device=DEV)
K.mask_pack(batch.input_mask.to(DEV), mb, B, L, Lp, out_t=mt)
sets = []
for _ in range(6):
    qkv = (torch.randn(B * L, 3 * H, device=DEV, generator=g) * 0.5).half()
    dctx = (torch.randn(B * L, H, device=DEV, generator=g) * 0.1).half()
    sets.append((qkv, dctx, torch.empty(B * L, H, device=DEV, dtype=torch.half), torch.empty(B, A, L, device=DEV), torch.empty_like(qkv), torch.empty(B, A, L, device=DEV)))
for q, d, c, l, dq, dl in sets:
    K.attn_fwd(q, mb, c, l, B, L, A, 0.125, dropout_p=0.1, seed=1)
for i in range(12):
    q, d, c, l, dq, dl = sets[i % 6]
    K.attn_bwd(q, mb, mt, c, d, l, dq, dl, B, L, A, 0.125, dropout_p=0.1, seed=1)
torch.cuda.synchronize()
lib = K.load(); buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.vlp_debug_read_attn_trace.argtypes = [C.c_void_p, C.c_int64]
rc = lib.vlp_debug_read_attn_trace(buf.ctypes.data, buf.nbytes); assert rc == 0, rc
t = buf.reshape(4096, 8)[:B * A].astype(np.int64)
t0 = t[:, 0].min()
names = ["loads issued+landed, LDS writes issued", "staging barrier", "pair 0 (wave 0)", "pairs 1.. (wave 0)", "dK/dV stores issued + phase barrier", "phase 2 (dQ tiles)"]
print("B=%d: workgroup start spread %d ticks; first end %d, last end %d ticks after the first start" % (B, t[:, 0].max() - t0, t[:, 6].min() - t0, t[:, 6].max() - t0))
for i, n in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print("%-44s median %7d  p10 %7d  p90 %7d ticks" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
d = t[:, 7] - t[:, 2]
for w in range(12):          # workgroup b recorded the end of phase 1 of its wave b % 12 (key tile w)
    dw = d[np.arange(len(d)) % 12 == w]
    print("phase 1 of the wave owning key tile %2d        median %7d  p10 %7d  p90 %7d ticks" % (w, np.median(dw), np.percentile(dw, 10), np.percentile(dw, 90)))
tot = t[:, 6] - t[:, 0]
print("%-44s median %7d  p10 %7d  p90 %7d ticks" % ("whole workgroup (wave 0)", np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
