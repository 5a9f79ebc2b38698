#!/usr/bin/env python3
"""Diagnostic: per-step cycle breakdown of the recurrent cell kernels (needs `make -C sparch_amd/csrc prof`
and SPARCH_HIP_LIB=sparch_amd/libsparch_hip_prof.so).  Prints average shader cycles per time step and segment
for wave 0 of every workgroup."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from sparch_amd import _capi  # noqa: E402
from sparch_amd import functional as Fn  # noqa: E402

B, T, H = 256, 250, 1024
kind = sys.argv[1] if len(sys.argv) > 1 else "RadLIF"
g = torch.Generator().manual_seed(0)
dev = "cuda"
V = torch.nn.init.orthogonal_(torch.empty(H, H), generator=g).to(dev)
Wx = (torch.randn(B, T, H, generator=g) * 1.2 + 0.2).to(dev).requires_grad_(True)
p = dict(alpha=torch.rand(H, generator=g) * 0.14 + 0.82, beta=torch.rand(H, generator=g) * 0.024 + 0.967,
         a=torch.rand(H, generator=g) * 2 - 1, b=torch.rand(H, generator=g) * 2)
p = {k: v.to(dev) for k, v in p.items()}
u0, w0, s0 = (torch.rand(B, H, generator=g).to(dev) for _ in range(3))
gs = torch.randn(B, T, H, generator=g).to(dev)
lib = _capi.lib
lib.sparch_rec_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros((2, 512, 12), np.uint64)


def run():
    s = Fn.SpikingCellFn.apply(kind, 1.0, Wx, p["alpha"], p["beta"], p["a"], p["b"], V, u0, w0, s0, None)
    (s * gs).sum().backward()
    torch.cuda.synchronize()
    return s


run()
lib.sparch_rec_prof_read(buf.ctypes.data, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
s = run()
e1.record()
torch.cuda.synchronize()
lib.sparch_rec_prof_read(buf.ctypes.data, 1)
print(f"fwd+bwd wall {e0.elapsed_time(e1):.3f} ms, firing rate {float(s.mean()):.4f}")
names = [["poll wait", "mfma+lds", "barrier", "pointwise+publish", "bulk stores", "-", "-", "-", "-", "-"],
         ["first load issue", "wait+split+mfma+lds", "barrier", "split+publish stores", "publish barrier", "stores+partial sums",
          "settle+prefetch issue", "tile reduction", "reverse-step arithmetic", "-"]]
if "allwaves" in sys.argv:  # library built with -DSPARCH_REC_PROF_ALLWAVES: one line per wave (mean over 64 workgroups)
    for w, label in [(0, "forward"), (1, "backward")]:
        a = buf[w, :, :10].astype(np.float64).reshape(64, 8, 10) / T
        print(f"{label}: cycles per step and wave (mean over 64 workgroups); columns: " + " | ".join(n for n in names[w] if n != "-"))
        for wave in range(8):
            print(f"   wave {wave}: " + " ".join(f"{a[:, wave, i].mean():7.0f}" for i in range(10) if names[w][i] != "-")
                  + f"   sum {a[:, wave].sum(1).mean():7.0f}")
    sys.exit(0)
for w, label in [(0, "forward"), (1, "backward"), (2, "forward, wave 4"), (3, "backward, wave 4")]:
    a = buf[w % 2, 256 * (w // 2):256 * (w // 2) + 256, :10].astype(np.float64) / T
    w = w % 2
    print(f"{label}: cycles per step (mean over workgroups | min | max)")
    for i in range(10):
        print(f"   {names[w][i]:20s} {a[:, i].mean():9.0f} | {a[:, i].min():9.0f} | {a[:, i].max():9.0f}")
    print(f"   {'sum':20s} {a.sum(1).mean():9.0f}  (= {a.sum(1).mean() / 100:.2f} us at 100 MHz s_memtime? see note)")
