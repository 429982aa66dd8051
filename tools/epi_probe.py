"""What does the FFN-up epilogue cost?  Same GEMM (10688 x 3072 x 768) with: bias only | + pre-activation store | + gelu | + both;
and the FFN-down dgrad form (gelu' multiply)."""
import sys, torch
sys.path.insert(0, ".")
from vlp_amd import _lib as K
DEV = torch.device("cuda:0")
M, N, Kd = 10688, 3072, 768
g = torch.Generator(device=DEV); g.manual_seed(0)
x = (torch.randn(M, Kd, device=DEV, generator=g) * 0.5).half(); w = (torch.randn(N, Kd, device=DEV, generator=g) * 0.05).half()
y = torch.empty(M, N, device=DEV, dtype=torch.half); z = torch.empty_like(y); bias = torch.zeros(N, device=DEV, dtype=torch.half)
zsrc = (torch.randn(M, N, device=DEV, generator=g)).half()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e3
for var in (2, 10, 12, 13, 5):
    r = {}
    r["bias"] = timeit(lambda: K.gemm_nt(x, w, y, M, N, Kd, bias=bias, variant=var))
    r["+preact"] = timeit(lambda: K.gemm_nt(x, w, y, M, N, Kd, bias=bias, preact=z, variant=var))
    r["+gelu"] = timeit(lambda: K.gemm_nt(x, w, y, M, N, Kd, bias=bias, act=K.ACT_GELU, variant=var))
    r["+relu"] = timeit(lambda: K.gemm_nt(x, w, y, M, N, Kd, bias=bias, act=K.ACT_RELU, variant=var))
    r["+both"] = timeit(lambda: K.gemm_nt(x, w, y, M, N, Kd, bias=bias, preact=z, act=K.ACT_GELU, variant=var))
    r["gelu'mul"] = timeit(lambda: K.gemm_nt(x, w, y, M, N, Kd, mul_src=zsrc, mul_mode=K.MUL_GELU_GRAD, variant=var))
    r["relu'mul"] = timeit(lambda: K.gemm_nt(x, w, y, M, N, Kd, mul_src=zsrc, mul_mode=K.MUL_RELU_MASK, variant=var))
    print("variant %2d: " % var + "  ".join("%s %.1f us" % kv for kv in r.items()))
