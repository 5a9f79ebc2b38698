"""Diagnostic: wall time of the headline GEMM shapes of a cfg3 step (B*T = 64000 rows, 1024 units), each
through functional.py as the layers call them.  SPARCH_HIP_LIB selects a library build to compare."""
import sys

import torch

sys.path.insert(0, ".")
from sparch_amd import functional as Fn  # noqa: E402

g = torch.Generator().manual_seed(0)
M = 64000
S = (torch.rand(M, 1024, generator=g) < 0.08).float().cuda()
S16 = S.to(torch.bfloat16)
D = torch.randn(M, 1024, generator=g).cuda()
X = torch.poisson(torch.full((M, 700), 0.05), generator=g).cuda()
W = torch.randn(1024, 1024, generator=g).cuda()
W0 = torch.randn(1024, 700, generator=g).cuda()
bias = torch.randn(1024, generator=g).cuda()
WP = Fn.split_planes(W)
xflag = Fn.flag_bf16_exact(X)


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3)


print("us:",
      "nt16+stats", t(lambda: Fn.gemm_nt(S, W, bias, colstat=True, spike_scale=1.0, a16=S16, b_planes=WP)),
      "nt_in(auto)", t(lambda: Fn.gemm_nt(X, W0, bias, colstat=True, a_exact_flag=xflag)),
      "nn6_wp", t(lambda: Fn.gemm_nn(D, W, b_planes=WP)),
      "nn6", t(lambda: Fn.gemm_nn(D, W)),
      "tn_dV(16)", t(lambda: Fn.gemm_tn(S16, D, zero_diag=True, spike_side=0, spike16=True)),
      "tn_dW(16)", t(lambda: Fn.gemm_tn(D, S16, spike_side=1, spike16=True)),
      "tn_dW0(auto)", t(lambda: Fn.gemm_tn(D, X, b_exact_flag=xflag)))
