#!/usr/bin/env python3
"""Diagnostic: achieved algorithmic HBM rate of the non-recurrent cell kernels (LIF / adLIF) vs problem size
(12 / 16 bytes per neuron-step each way)."""
import sys

import torch

sys.path.insert(0, ".")
from sparch_amd import functional as Fn  # noqa: E402

dev = "cuda"
for kind, per in (("LIF", 12), ("adLIF", 16)):
    for B, T, H in ((128, 250, 512), (256, 250, 1024), (512, 250, 2048)):
        g = torch.Generator().manual_seed(0)
        Wx = (torch.randn(B, T, H, generator=g) * 1.2 + 0.2).to(dev).requires_grad_(True)
        p = dict(alpha=torch.rand(H, generator=g) * 0.14 + 0.82, beta=torch.rand(H, generator=g) * 0.024 + 0.967,
                 a=torch.rand(H, generator=g) * 2 - 1, b=torch.rand(H, generator=g) * 2)
        p = {k: v.to(dev) for k, v in p.items()}
        u0, w0, s0 = (torch.rand(B, H, generator=g).to(dev) for _ in range(3))
        gs = torch.randn(B, T, H, generator=g).to(dev)
        ad = kind == "adLIF"
        Fn.timer.enabled = True
        for it in range(4):
            if it == 1:
                Fn.timer.reset()
            s = Fn.SpikingCellFn.apply(kind, 1.0, Wx, p["alpha"], p["beta"] if ad else None, p["a"] if ad else None,
                                       p["b"] if ad else None, None, u0, w0 if ad else None, s0, None)
            (s * gs).sum().backward()
        torch.cuda.synchronize()
        tot = Fn.timer.collect()
        line = f"{kind:6s} B={B:4d} T={T} H={H:5d}:"
        for k, (c, t) in tot.items():
            if "cell" in k:
                line += f"  {k} {t / c:.3f} ms = {per * B * T * H / (t / c * 1e-3) / 1e12:.2f} TB/s"
        print(line)
