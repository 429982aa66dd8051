"""Run attn_fwd from the trace build and print per-phase cycle statistics (VLP_HIP_LIB must point at libvlp_hip_trace.so)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K
from vlp_amd import synthetic as S
DEV = torch.device("cuda:0"); B, L, A, H = 64, 167, 12, 768
g = torch.Generator(device=DEV); g.manual_seed(0)
batch = S.make_batch(B, max_len_b=64, vocab_size=1000, max_pred=3, s2s_prob=1.0, seed=3)
Lp = 192
mb = torch.empty(B, L, Lp, dtype=torch.uint8, device=DEV); K.mask_pack(batch.input_mask.to(DEV), mb, B, L, Lp)
sets = [((torch.randn(B * L, 3 * H, device=DEV, generator=g) * 0.5).half(), torch.empty(B * L, H, device=DEV, dtype=torch.half), torch.empty(B, A, L, device=DEV)) for _ in range(6)]
for i in range(12):
    q, c, l = sets[i % 6]
    K.attn_fwd(q, mb, c, l, B, L, A, 0.125, dropout_p=0.1, seed=1)
torch.cuda.synchronize()
lib = K.load(); buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.vlp_debug_read_attn_trace.argtypes = [C.c_void_p, C.c_int64]
rc = lib.vlp_debug_read_attn_trace(buf.ctypes.data, buf.nbytes); assert rc == 0, rc
t = buf.reshape(4096, 8)[:B * A].astype(np.int64)
names = ["stage issue (global->reg->LDS writes)", "wait barrier (staging lands)", "first tile: Q + mask arrive", "S MFMAs (+ live test)", "scale + mask + row max", "exp2 + dropout + PV + store", "remaining tiles of wave 0"]
t0 = t[:, 0].min()
print("workgroup start spread: %.1f us; end spread: first %.1f us last %.1f us (100 MHz counter? ticks shown raw)" % ((t[:, 0].max() - t0) / 100.0, (t[:, 7].min() - t0) / 100.0, (t[:, 7].max() - t0) / 100.0))
for i, n in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print("%-42s median %7d  p10 %7d  p90 %7d ticks" % (n, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
tot = t[:, 7] - t[:, 0]
print("%-42s median %7d  p10 %7d  p90 %7d ticks" % ("whole workgroup (wave 0)", np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
order = np.argsort(t[:, 0]); print("start ticks of workgroups (sorted, every 64th):", ((t[order, 0] - t0)[::64]).tolist())
