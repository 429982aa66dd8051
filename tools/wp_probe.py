"""Times vlp_gemm_nt variants on the training shapes with cold operands (12 rotating operand sets); used with the investigation
builds of tools/build_wp_dbg.sh:  VLP_HIP_LIB=vlp_amd/libvlp_hip_wpd6.so python tools/wp_probe.py 76,77 768x3072,3072x768"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlp_amd import _lib as K
DEV = torch.device("cuda:0")
def timeit(fn, iters=36, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
variants = [int(v) for v in sys.argv[1].split(",")]
shapes = [tuple(int(a) for a in s.split("x")) for s in sys.argv[2].split(",")]
M = int(os.environ.get("M", "10688")); ROT = int(os.environ.get("ROT", "12"))
tag = os.path.basename(os.environ.get("VLP_HIP_LIB", "product"))
for N, Kd in shapes:
    xs = [(torch.randn(M, Kd, device=DEV) * 0.5).half() for _ in range(ROT)]
    ws = [(torch.randn(N, Kd, device=DEV) * 0.05).half() for _ in range(ROT)]
    y = torch.empty(M, N, device=DEV, dtype=torch.float16)
    row = []
    for v in variants:
        i = [0]
        def f():
            K.gemm_nt(xs[i[0] % ROT], ws[i[0] % ROT], y, M, N, Kd, variant=v)
            i[0] += 1
        row.append("v%d:%.1f" % (v, timeit(f)))
    print("%-28s M=%d N=%d K=%d rot=%d | %s" % (tag, M, N, Kd, ROT, " ".join(row)), flush=True)
