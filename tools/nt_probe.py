import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K
DEV = torch.device("cuda:0")
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
N, Kd, v = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for M in (10688, 5376, 2688, 1344):
    ROT = int(os.environ.get('ROT', '8'))
    PAD = int(os.environ.get("PAD", "0"))
    xs = [(torch.randn(M, Kd + PAD, device=DEV) * 0.5).half() for _ in range(ROT)]
    ws = [(torch.randn(N, Kd + PAD, device=DEV) * 0.5).half() for _ in range(ROT)]
    y = torch.empty(M, N, device=DEV, dtype=torch.float16)
    i = [0]
    def f():
        K.gemm_nt(xs[i[0] % ROT], ws[i[0] % ROT], y, M, N, Kd, variant=v)
        i[0] += 1
    us = timeit(f)
    tiles = ((M + 255) // 256) * ((N + (127 if v == 27 else 255)) // (128 if v == 27 else 256))
    print("rot=%d pad=%s dbg=%s M=%5d N=%d K=%d v%d  WGs=%4d  %.1f us" % (ROT, os.environ.get("PAD", "0"), os.environ.get("VLP_NT_DEBUG", "0"), M, N, Kd, v, tiles, us))
