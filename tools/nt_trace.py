"""Per-workgroup phase times of one NT GEMM from the investigation build (VLP_HIP_LIB=vlp_amd/libvlp_hip_dbg.so, built with
tools/build_variant_lib.sh vlp_amd/libvlp_hip_dbg.so -DVLP_NT_DEBUG).   python tools/nt_trace.py N K variant [sg]"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K
DEV = torch.device("cuda:0"); M = 64 * 167
N, Kd, v = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sg = len(sys.argv) > 4
ROT = 8
xs = [(torch.randn(M, Kd, device=DEV) * 0.5).half() for _ in range(ROT)]
ws = [(torch.randn(N, Kd, device=DEV) * 0.5).half() for _ in range(ROT)]
ys = [torch.empty(M, N, device=DEV, dtype=torch.half) for _ in range(ROT)]
ps = [torch.empty(M, N, device=DEV, dtype=torch.half) for _ in range(ROT)]
bias = torch.zeros(N, device=DEV, dtype=torch.half)
for i in range(ROT + 3):
    j = i % ROT
    if sg:
        K.gemm_nt(xs[j], ws[j], ys[j], M, N, Kd, bias=bias, preact=ps[j], act=K.ACT_GELU_SAVE_GRAD, variant=v)
    else:
        K.gemm_nt(xs[j], ws[j], ys[j], M, N, Kd, variant=v)
torch.cuda.synchronize()
lib = K.load(); buf = np.zeros(4096 * 4, dtype=np.uint64)
lib.vlp_debug_read_nt_trace.argtypes = [C.c_void_p, C.c_int64]
rc = lib.vlp_debug_read_nt_trace(buf.ctypes.data, buf.nbytes); assert rc == 0, rc
bm, bn = (256, 256) if (v & 7) == 5 else ((256, 128) if (v & 7) == 3 else (128, 128))
nwg = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
t = buf.reshape(4096, 4)[:nwg].astype(np.int64)
t0 = t[:, 0].min()
us = lambda x: x / 100.0
print("N=%d K=%d variant %d%s: %d workgroups; kernel span %.1f us (first start -> last store acknowledged)" % (N, Kd, v, " savegrad" if sg else "", nwg, us(t[:, 3].max() - t0)))
start = t[:, 0] - t0
first = start < np.percentile(start, 45)        # round 1 = the workgroups that start at once
for name, sel in (("round 1", first), ("later rounds", ~first)):
    if not sel.any():
        continue
    d = t[sel]
    print("  %-12s n=%4d  start %5.1f..%5.1f us | main loop %5.1f (p10 %5.1f p90 %5.1f) | epilogue issue %5.1f (p90 %5.1f) | stores acknowledged +%5.1f (p90 %5.1f) us" % (
        name, sel.sum(), us(d[:, 0].min() - t0), us(d[:, 0].max() - t0), us(np.median(d[:, 1] - d[:, 0])), us(np.percentile(d[:, 1] - d[:, 0], 10)),
        us(np.percentile(d[:, 1] - d[:, 0], 90)), us(np.median(d[:, 2] - d[:, 1])), us(np.percentile(d[:, 2] - d[:, 1], 90)),
        us(np.median(d[:, 3] - d[:, 2])), us(np.percentile(d[:, 3] - d[:, 2], 90))))
