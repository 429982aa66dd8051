import sys, torch
sys.path.insert(0, ".")
from vlp_amd import _lib as K
DEV = torch.device("cuda:0")
M, N, Kd = 10688, 3072, 768
g = torch.Generator(device=DEV); g.manual_seed(0)
a = (torch.randn(M, N, device=DEV, generator=g) * 0.5).half()
b = (torch.randn(M, Kd, device=DEV, generator=g) * 0.5).half()
c = torch.empty(N, Kd, device=DEV, dtype=torch.half)
ws = torch.empty(K.gemm_tn_workspace_bytes(M, N, Kd), device=DEV, dtype=torch.uint8)
x = (torch.randn(M, Kd, device=DEV, generator=g) * 0.5).half()
w = (torch.randn(N, Kd, device=DEV, generator=g) * 0.5).half()
y = torch.empty(M, N, device=DEV, dtype=torch.half)
for var in (2, 26):
    for _ in range(3):
        K.gemm_tn(a, b, c, M, N, Kd, workspace=ws, variant=var, splits=8)
for var in (1, 2, 4, 12):
    for _ in range(3):
        K.gemm_nt(x, w, y, M, N, Kd, variant=var)
torch.cuda.synchronize()
