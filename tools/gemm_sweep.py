"""Diagnostic: wall time of the headline TN GEMM shapes (spike x dense both ways, dense x dense)."""
import sys, torch
sys.path.insert(0, ".")
from sparch_amd import functional as Fn
g = torch.Generator().manual_seed(0)
M = 64000
S = (torch.rand(M, 1024, generator=g) < 0.08).float().cuda()
D = torch.randn(M, 1024, generator=g).cuda()
X = torch.randn(M, 700, generator=g).cuda()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("dV", round(t(lambda: Fn.gemm_tn(S, D, zero_diag=True, spike_side=0)), 4),
      "dW1", round(t(lambda: Fn.gemm_tn(D, S, spike_side=1)), 4),
      "dW0(6)", round(t(lambda: Fn.gemm_tn(D, X)), 4))
