#!/usr/bin/env python3
"""Diagnostic: phase breakdown of one K-tile iteration of the split GEMM kernels (needs `make prof` and
SPARCH_HIP_LIB=sparch_amd/libsparch_hip_prof.so).  Stamps come from workgroup 7, wave 0."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from sparch_amd import _capi  # noqa: E402
from sparch_amd import functional as Fn  # noqa: E402

lib = _capi.lib
lib.sparch_gemm_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(24, np.uint64)
g = torch.Generator().manual_seed(0)
M = 64000
S = (torch.rand(M, 1024, generator=g) < 0.08).float().cuda()
D = torch.randn(M, 1024, generator=g).cuda()
W = torch.randn(1024, 1024, generator=g).cuda()
S16 = S.to(torch.bfloat16)
bias = torch.randn(1024, generator=g).cuda()
WP = Fn.split_planes(W)
cases = {
    "spike16_nt + BN stats (hidden projection)": (lambda: Fn.gemm_nt(S, W, bias, colstat=True, spike_scale=1.0, a16=S16), 32),
    "spike16_nt_wp + BN stats": (lambda: Fn.gemm_nt(S, W, bias, colstat=True, spike_scale=1.0, a16=S16, b_planes=WP), 32),
    "gemm6_nn_wp dX": (lambda: Fn.gemm_nn(D, W, b_planes=WP), 32),
    "spike_tn dV (E on A)": (lambda: Fn.gemm_tn(S, D, zero_diag=True, spike_side=0), 2000 / 16),
    "spike_tn dW (E on B)": (lambda: Fn.gemm_tn(D, S, spike_side=1), 2000 / 16),
    "spike_nt W1 fwd": (lambda: Fn.gemm_nt(S, W, spike_scale=1.0), 32),
    "gemm6_nn ds0": (lambda: Fn.gemm_nn(D, W), 32),
}
names = ["wait+convert+LDS write", "barrier 1", "issue loads", "frag reads + MFMA", "barrier 2"]
for label, (fn, tiles) in cases.items():
    fn(); torch.cuda.synchronize()
    lib.sparch_gemm_prof_read(buf.ctypes.data, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    lib.sparch_gemm_prof_read(buf.ctypes.data, 1)
    per = buf[:5].astype(np.float64) / tiles
    print(f"{label}: {e0.elapsed_time(e1):.3f} ms; cycles per K-tile: " +
          ", ".join(f"{n} {v:.0f}" for n, v in zip(names, per)) + f"  | sum {per.sum():.0f}"
          f" | workgroup lifetime {int(buf[5])} s_memtime ticks = {int(buf[6]) / 100:.1f} us"
          f" -> s_memtime runs at {float(buf[5]) / max(float(buf[6]), 1) * 100:.0f} MHz")
    print("    per wave (phase | barrier wait) cycles per K-tile: " +
          "  ".join(f"w{w}: {buf[8 + 2 * w] / tiles:.0f}|{buf[9 + 2 * w] / tiles:.0f}" for w in range(8)))
