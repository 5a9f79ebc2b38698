"""Is the dense six-term product power-bound?  The same kernel, the same instruction stream, operands of different
switching activity: random fp32 values (all three planes of both operands busy), the same values rounded to bf16
(mid / lo planes all zero: four of the six MFMA terms multiply zeros), and all-zero operands.  Prints us per product
[64000 x 1024 x 1024] (HIP events, 20 launches)."""
import sys

import torch

sys.path.insert(0, ".")
from sparch_amd import functional as Fn  # noqa: E402

g = torch.Generator().manual_seed(0)
M = 64000
D = torch.randn(M, 1024, generator=g).cuda()
W = torch.randn(1024, 1024, generator=g).cuda()


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3)


cases = {"random fp32": (D, W), "bf16-representable": (D.bfloat16().float(), W.bfloat16().float()),
         "zeros": (torch.zeros_like(D), torch.zeros_like(W))}
for rep in range(2):
    print("us:", "  ".join(f"{name}: {t(lambda a=a, b=b: Fn.gemm_nn(a, b))}" for name, (a, b) in cases.items()), flush=True)
