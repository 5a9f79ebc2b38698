import sys, torch
sys.path.insert(0, ".")
from sparch_amd import functional as Fn
g = torch.Generator().manual_seed(0)
M = 64000
S = (torch.rand(M, 1024, generator=g) < 0.08).float().cuda()
D = torch.randn(M, 1024, generator=g).cuda()
for _ in range(3):
    Fn.gemm_tn(S, D, zero_diag=True, spike_side=0)
torch.cuda.synchronize()
