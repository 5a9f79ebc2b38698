#!/usr/bin/env python3
"""Times the recurrent cell kernels alone (B=256, T=250, H=1024 by default) with HIP events: ms per launch, forward
and backward.  SPARCH_HIP_LIB selects a diagnostic build.  Usage: tools/rec_time.py [kind] [B] [T] [H]"""
import sys

import torch

sys.path.insert(0, ".")
from sparch_amd import functional as Fn  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "RadLIF"
B, T, H = (int(v) for v in (sys.argv[2:5] + ["256", "250", "1024"][len(sys.argv[2:5]):]))
g = torch.Generator().manual_seed(0)
dev = "cuda"
V = torch.nn.init.orthogonal_(torch.empty(H, H), generator=g).to(dev)
Wx = (torch.randn(B, T, H, generator=g) * 1.2 + 0.2).to(dev).requires_grad_(True)
p = dict(alpha=torch.rand(H, generator=g) * 0.14 + 0.82, beta=torch.rand(H, generator=g) * 0.024 + 0.967,
         a=torch.rand(H, generator=g) * 2 - 1, b=torch.rand(H, generator=g) * 2)
p = {k: v.to(dev) for k, v in p.items()}
u0, w0, s0 = (torch.rand(B, H, generator=g).to(dev) for _ in range(3))
gs = torch.randn(B, T, H, generator=g).to(dev)
adaptive = kind in ("adLIF", "RadLIF")
Fn.timer.enabled = True
import os
N_IT = int(os.environ.get("REC_TIME_ITERS", "40"))
for it in range(N_IT):
    if it == 4:
        Fn.timer.collect()
        Fn.timer.reset()
    s = Fn.SpikingCellFn.apply(kind, 1.0, Wx, p["alpha"], p["beta"] if adaptive else None, p["a"] if adaptive else None,
                               p["b"] if adaptive else None, V, u0, w0 if adaptive else None, s0, None)
    (s * gs).sum().backward()
tot = Fn.timer.collect()
Fn.check_status()
print(" ".join(f"{k}: {v[1] / v[0]:.4f} ms" for k, v in tot.items() if "rec" in k or "cell" in k), flush=True)
