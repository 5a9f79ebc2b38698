"""
Host orchestration of the HIP hot path: one torch.autograd.Function per layer type.

Each Function replaces the eager op chain of a reference layer's forward
(snns.py:249-280 / 386-417 / 521-552 / 663-694 for the spiking layers, 793-806 for the
readout) and the autograd tape replay behind `loss.backward()` (exp.py:376) with a fixed
sequence of libsparch_hip.so calls on the current HIP stream.  PyTorch here only owns
device memory, the stream and the autograd graph edges between layers.

No CPU fallback: CPU tensors raise.
"""
import ctypes
import os

import torch

from . import _capi
from ._capi import KIND, check, lib, ptr

# Exact bf16-split MFMA GEMMs for spike operands (SPARCH_SPIKE_GEMM=0 forces the fp32 MFMA everywhere).
USE_SPIKE_GEMM = os.environ.get("SPARCH_SPIKE_GEMM", "1") != "0"
# spike operands travel between layers as bf16 0/1 planes next to the fp32 tensors (half the GEMM operand bytes)
USE_SPIKE16 = os.environ.get("SPARCH_SPIKE16", "1") != "0"
USE_PRESPLIT = os.environ.get("SPARCH_PRESPLIT", "1") != "0"  # weights split into bf16 planes once per step
PRESPLIT_NT = os.environ.get("SPARCH_PRESPLIT_NT", "0") == "1"  # ... also for the forward projection (see SpikingLayerFn)

# Dense GEMMs: "split6" = exact 6-term bf16 split on the bf16 MFMA (default), "fp32" = fp32-input MFMA.
DENSE_GEMM = os.environ.get("SPARCH_DENSE_GEMM", "split6")

# The gradient of a BatchNorm'd projection may leave the BatchNorm pass as its three exact bf16 planes (made once) so
# that the dX / dW products read planes instead of re-splitting the fp32 tensor in every workgroup that stages a tile
# of it; results are bit-identical either way.  Measured (round 3, A/B in one call): at the headline shape the
# products do not get faster for it (dX 0.69 -> 0.67 ms, dW unchanged: conversion VALU is NOT what bounds them) and
# the pass writes half as much again (0.125 -> 0.165 ms): 6.58 -> 6.63 ms per step.  For a BIDIRECTIONAL layer the
# same pass also adds the two directions' gradients (no separate sparch_add_halves pass): cfg5 76.1 -> 74.7 ms.
# Hence "auto" = bidirectional layers only; SPARCH_DX_PLANES=1 / 0 forces it on / off.
USE_DX_PLANES = {"1": True, "0": False}.get(os.environ.get("SPARCH_DX_PLANES", "auto"), "auto")

# BatchNorm backward's column sums (dbeta, dgamma) come out of the cell's backward kernel instead of a
# separate pass over dy and x (SPARCH_FUSE_BN_SUMS=0: the separate sparch_bn_bwd_reduce pass, for comparison).
FUSE_BN_SUMS = os.environ.get("SPARCH_FUSE_BN_SUMS", "1") != "0"

# Saved states (u, w) of the spiking layers in bf16 instead of fp32 (SPARCH_SAVE_DTYPE=bf16; BASELINE configs[4]
# is the long-sequence bf16 case: at T=1000 the fp32 saves are 4.2 GB per layer and direction pair).  Every
# discrete decision of the backward pass stays exactly the fp32 one (csrc/common.h save_u16), so spikes, dWx, dW
# and dV do not change; dalpha / dbeta / da see the 2^-9 rounding of u and w: stated tolerance 2e-2 of max-abs
# against the fp32 path (tests/test_hip_parity.py::test_bf16_saved_states_*).  Off by default.
SAVE_BF16 = os.environ.get("SPARCH_SAVE_DTYPE", "fp32").lower() == "bf16"

# Operand precision of every matrix product (include/sparch_hip.h sparch_set_operand_precision): "fp32" = exact
# bf16 splits of fp32 operands (default, fp32 results), "bf16" = operands rounded once to bf16, fp32
# accumulation, fp32 states / statistics / parameter updates (BASELINE configs[4]; SPARCH_COMPUTE_DTYPE=bf16 or
# run_exp.py --compute_dtype bf16).  Stated tolerance against the fp32 path: tests/test_hip_parity.py
# ::test_bf16_operand_mode_*.
COMPUTE_DTYPES = {"fp32": 0, "bf16": 1}


_precision = 0  # what this module passes as the `precision` argument of every matrix-product call (the C library
#                keeps no such state since ABI v5: include/sparch_hip.h)


def set_compute_dtype(name):
    """Select the operand precision this module asks of the library's matrix products; returns the previous
    setting's name."""
    global _precision
    name = {"float32": "fp32", "f32": "fp32", "bfloat16": "bf16"}.get(str(name).lower(), str(name).lower())
    if name not in COMPUTE_DTYPES:
        raise ValueError(f"compute dtype must be one of {sorted(COMPUTE_DTYPES)}, got {name!r}")
    prev = compute_dtype()
    _precision = COMPUTE_DTYPES[name]
    return prev


def compute_dtype():
    return "bf16" if _precision == 1 else "fp32"


def _prec():
    return _precision


if os.environ.get("SPARCH_COMPUTE_DTYPE"):
    set_compute_dtype(os.environ["SPARCH_COMPUTE_DTYPE"])

SEED_IN_MEMORY = 1 << 63  # include/sparch_hip.h SPARCH_SEED_IN_MEMORY: the seed argument carries a device address

BN_MOMENTUM = 0.05  # snns.py:240
# SyncBN for data-parallel runs (off by default = standard DDP semantics: per-rank statistics).
# {"group": process group or None, "world": n}: BatchNorm then normalises with the statistics of the GLOBAL
# batch — forward: every rank's per-tile (sum x, sum x^2) partials are all-gathered and finished in one pass
# (same fp64 reduction as the single-device path, so world ranks x B/world rows reproduce one device with B
# rows up to summation order); backward: (sum dy, sum dy*xhat) are all-reduced for dx, the affine
# parameters' own gradients stay local sums (the gradient all-reduce averages them like every other one).
SYNC_BN = None
NORM_EPS = 1e-5

ALPHA_LIM = (0.8187307530779818, 0.9607894391523232)  # exp(-1/5), exp(-1/25)   snns.py:229
BETA_LIM = (0.9672161004820059, 0.9917012926388759)  # exp(-1/30), exp(-1/120)  snns.py:357
A_LIM = (-1.0, 1.0)  # snns.py:358
B_LIM = (0.0, 2.0)  # snns.py:359

_status = {}

# Objects with pre_persistent() / post_persistent(), called right before / right behind every persistent recurrent
# launch (kernels whose workgroups wait for each other and need the whole GPU): sparch_amd.dp's "window" policy
# keeps its collectives out of those kernels' way with them.
persistent_hooks = []


class _persistent_launch:
    """with _persistent_launch(): <enqueue one persistent kernel>"""

    def __enter__(self):
        for h in persistent_hooks:
            h.pre_persistent()

    def __exit__(self, *exc):
        if exc[0] is None:
            for h in persistent_hooks:
                h.post_persistent()
        return False



class KernelTimer:
    """Optional HIP-event timing of named C-ABI calls on the current stream (bench.py uses it for
    the roofline object).  Disabled by default: zero overhead on the training path."""

    def __init__(self):
        self.enabled = False
        self.pending = []   # (name, start_event, end_event)
        self.totals = {}    # name -> [count, total_ms]

    def start(self, name):
        if not self.enabled:
            return None
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        return (name, a, b)

    def stop(self, tok):
        if tok is not None:
            tok[2].record()
            self.pending.append(tok)

    def collect(self):
        torch.cuda.synchronize()
        for name, a, b in self.pending:
            c = self.totals.setdefault(name, [0, 0.0])
            c[0] += 1
            c[1] += a.elapsed_time(b)
        self.pending = []
        return self.totals

    def reset(self):
        self.pending, self.totals = [], {}


timer = KernelTimer()


def _require_device(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            f"sparch_amd: {what} is on '{t.device}'. The MI355X path has no CPU fallback; "
            "move the model and inputs to a HIP device (.to('cuda'))."
        )


def _stream():
    return torch.cuda.current_stream().cuda_stream


def status_word(device):
    """Per-device uint32 raised by a recurrent kernel whose in-kernel wait timed out."""
    key = torch.device(device).index or 0
    if key not in _status:
        _status[key] = torch.zeros(4, dtype=torch.int32, device=device)
    return _status[key]


_TIMEOUT_TEXT = (
    "recurrent cell kernel: in-kernel wait timed out (SPARCH_ETIMEOUT).  The persistent kernels need one "
    "workgroup per CU resident at the same time; if the GPU is shared with another process (or partitioned) "
    "they cannot all become resident.  The results of the affected step are invalid (the optimizer step and "
    "the BatchNorm running statistics skip themselves on the device while the status word is raised).")
_degraded = set()  # device indices whose recurrent kernels now run one launch per time step


STATUS_KERNELS = {0: "?", 1: "rec_fwd (RLIF / RadLIF forward)", 2: "rec_bwd (RLIF / RadLIF backward)",
                  3: "ann_rec forward (RNN)", 4: "ann_rec backward (RNN)", 5: "ligru forward", 6: "ligru backward",
                  7: "gru forward", 8: "gru backward"}  # include/sparch_hip.h SPARCH_STATUS_*
last_timeout = None   # details of the most recent timeout poll_status() saw: dict(kernel, step, skipped_steps)
_timeout_listeners = []  # callables(info) — e.g. the optimizer takes its step counter back by info["skipped_steps"]


def poll_status(device="cuda"):
    """Synchronising read of the persistent-kernel status word (4 x uint32: raised, optimizer steps skipped since,
    kernel id, time step); clears it.  True = a wait timed out; the details are left in `last_timeout` and handed
    to the registered listeners."""
    global last_timeout
    w = status_word(device)
    vals = w.tolist()
    if vals[0] != 0:
        w.zero_()
        step = vals[3] & 0xFFFFFFFF
        last_timeout = {"kernel": STATUS_KERNELS.get(vals[2], f"kernel id {vals[2]}"),
                        "step": None if step == 0xFFFFFFFF else step, "skipped_steps": int(vals[1]),
                        "device": str(w.device)}
        for fn in list(_timeout_listeners):
            fn(last_timeout)
        return True
    return False


def describe_timeout():
    """One line about the most recent timeout (which kernel, which time step, how many optimizer steps were skipped)."""
    if last_timeout is None:
        return "no timeout recorded"
    t = last_timeout
    import os as _os
    rank = _os.environ.get("RANK")
    return (f"{'rank ' + rank + ', ' if rank is not None else ''}{t['device']}: {t['kernel']} gave up"
            f"{'' if t['step'] is None else ' at time step ' + str(t['step'])}; "
            f"{t['skipped_steps']} optimizer step(s) were skipped on the device since")


def degrade(device="cuda"):
    """From now on this process runs the recurrent cells with one launch per time step on `device` (nothing
    waits inside a launch, any CU share works).  New launches only — the process is never re-executed."""
    _degraded.add(torch.device(device).index or 0)


def check_status(device="cuda", on_timeout="raise"):
    """Synchronising check of the persistent-kernel status word (call at a sync point).
    on_timeout = "raise": SparchHipError.  "degrade": switch this process to one launch per time step
    (see `degrade`) and return True, so that a trainer can log the lost step and carry on."""
    if not poll_status(device):
        return False
    if on_timeout == "degrade":
        degrade(device)
        return True
    raise _capi.SparchHipError(_TIMEOUT_TEXT + "  [" + describe_timeout() + "]  Set SPARCH_REC_STEPS_PER_LAUNCH=1 "
                               "(one launch per time step, no waiting inside) to run on a shared GPU.")


def rec_steps_per_launch(T):
    """Time steps per persistent launch of the recurrent cell kernels (default: whole sequence; 1 after a
    timeout put the current device into degraded mode)."""
    v = os.environ.get("SPARCH_REC_STEPS_PER_LAUNCH", "")
    if v:
        return int(v)
    if _degraded and torch.cuda.is_available() and torch.cuda.current_device() in _degraded:
        return 1
    return T


# Hidden sizes above this take the step path of the recurrent cells (the persistent kernels keep a (H x 32)
# slice of V in registers, which stops fitting at H > 1024); SPARCH_REC_STEP_PATH=1 forces it at any size (tests).
REC_PERSISTENT_MAX_H = 1024


def rec_step_path(H):
    return H > REC_PERSISTENT_MAX_H or os.environ.get("SPARCH_REC_STEP_PATH", "0") == "1"


def ligru_persistent_ok(H):
    """The LiGRU persistent kernels (gatedcell.hip) take hidden sizes that are multiples of 32 up to 1024;
    SPARCH_LIGRU_PERSISTENT=0 forces the launch-per-step path (comparison in tests)."""
    return H % 32 == 0 and H <= 1024 and os.environ.get("SPARCH_LIGRU_PERSISTENT", "1") != "0"


def gru_persistent_ok(H):
    """The GRU persistent kernels (gatedcell.hip): hidden sizes that are multiples of 32 up to 1024 whose H / 16
    workgroups per row tile fit the GPU at once (two hand-offs inside a step: no per-step degenerate form, so a
    device in degraded mode takes the launch-per-step path); SPARCH_GRU_PERSISTENT=0 forces that path."""
    if H % 32 != 0 or H > 1024 or os.environ.get("SPARCH_GRU_PERSISTENT", "1") == "0":
        return False
    if _degraded and torch.cuda.is_available() and torch.cuda.current_device() in _degraded:
        return False
    return H // 16 <= lib.sparch_device_cus()


def _f32c(t):
    return t.contiguous().float() if (t.dtype != torch.float32 or not t.is_contiguous()) else t


# ----------------------------------------------------------------------------- primitives
def flag_bf16_exact(x):
    """Device uint32: 1 iff every element of x is exactly representable in bf16 (no host sync)."""
    flag = torch.empty(4, dtype=torch.int32, device=x.device)
    check(lib.sparch_flag_bf16_exact(x.numel(), ptr(x), ptr(flag), _stream()), "sparch_flag_bf16_exact")
    return flag


USE_INPUT_PLANE = os.environ.get("SPARCH_INPUT_PLANE", "1") != "0"


def plane_bf16_exact(x2):
    """(plane, flag) of a (M, K) fp32 network input: the bf16 plane of x (rows padded to a multiple of 8 elements,
    zeros behind column K) and the device flag "every element is bf16-exact", made in ONE pass — the first
    layer's GEMMs then read the plane (2 bytes per element) when the flag is 1 and x itself otherwise."""
    M, K = x2.shape
    ldp = (K + 7) // 8 * 8
    plane = torch.empty(M, ldp, dtype=torch.bfloat16, device=x2.device)
    flag = torch.empty(4, dtype=torch.int32, device=x2.device)
    check(lib.sparch_plane_bf16_exact(M, K, ptr(x2), x2.stride(0), ptr(plane), ldp, ptr(flag), _stream()),
          "sparch_plane_bf16_exact")
    return plane, flag


_flag_one = {}


def input_from_counts(counts):
    """A batch of binned spike counts as (B,T,K) uint8 ON THE DEVICE -> the network input for SNN.forward: a
    (B,T,K) fp32 PLACEHOLDER (no values are written) carrying the bf16 plane the first layer's GEMMs read, made by
    one pass over the bytes (`sparch_expand_counts_u8`).  The reference uploads the dense float batch every step
    (exp.py:355-356: 179 MB at the headline shape against 45 MB of bytes); counts up to 255 are exact in both
    uint8 and bf16, so the plane — and everything computed from it — is the one `plane_bf16_exact` makes from the
    float batch."""
    _require_device(counts, "counts")
    if counts.dtype != torch.uint8 or counts.ndim != 3:
        raise ValueError("input_from_counts: a (B,T,K) uint8 tensor of spike counts")
    counts = counts.contiguous()
    B, T, K = counts.shape
    M, ldp = B * T, (K + 7) // 8 * 8
    plane = torch.empty(M, ldp, dtype=torch.bfloat16, device=counts.device)
    check(lib.sparch_expand_counts_u8(M, K, ptr(counts), ptr(plane), ldp, None, 0, _stream()), "sparch_expand_counts_u8")
    key = str(counts.device)
    one = _flag_one.get(key)
    if one is None:
        one = _flag_one[key] = torch.ones(4, dtype=torch.int32, device=counts.device)
    x = spike_placeholder(B, T, K, counts.device)
    x = x.view(B, T, K)  # a tensor object of its own for the tag (same one-element storage)
    x._sparch_input_plane = (tuple(x.shape), plane, one)
    return x


def input_plane_of(x):
    """(plane, flag) if x is an input made by `input_from_counts`, else None."""
    tag = getattr(x, "_sparch_input_plane", None)
    if tag is None or tag[0] != tuple(x.shape):
        return None
    return tag[1], tag[2]


def split_planes(W):
    """The three exact bf16 planes of a weight matrix, (3, *W.shape) bf16 (W = p0 + p1 + p2 exactly, the
    truncation split the GEMM kernels otherwise redo in every workgroup that stages a tile of W); None where
    the pre-split kernels do not apply (SPARCH_PRESPLIT=0, or a layout they do not take)."""
    if not USE_PRESPLIT or W.dtype != torch.float32 or not W.is_contiguous() or W.numel() % 8 or W.shape[-1] % 8:
        return None
    planes = torch.empty((3,) + tuple(W.shape), dtype=torch.bfloat16, device=W.device)
    check(lib.sparch_split3(W.numel(), ptr(W), ptr(planes), _stream()), "sparch_split3")
    return planes


def gemm_nt(A, B, bias=None, colstat=False, spike_scale=None, a_exact_flag=None, a16=None, b_planes=None,
            a_plane=None):
    """A (M,K) @ B (N,K)^T (+bias) -> (M,N); optional BatchNorm column-stat partials.
    spike_scale = c: A's entries are 0 or c (a spike train) -> exact bf16-split MFMA path;
    a16: the same spikes as a (M,K) bf16 0/1 plane (read instead of A);
    b_planes: split_planes(B) (same result; the kernel copies the planes instead of converting B)."""
    M, K = A.shape
    N = B.shape[0]
    C = torch.empty(M, N, dtype=torch.float32, device=A.device)
    ws = None
    if colstat:
        ws = torch.empty(2 * ((M + 127) // 128) * N, dtype=torch.float32, device=A.device)
    if spike_scale is not None and USE_SPIKE_GEMM and a16 is not None and USE_SPIKE16 and b_planes is not None:
        tok = timer.start(f"gemm_spike_nt[{M}x{N}x{K}]")
        check(lib.sparch_gemm_spike16_nt_wp(M, N, K, ptr(a16), a16.stride(0), float(spike_scale), ptr(B),
                                            ptr(b_planes), B.stride(0), ptr(C), N, ptr(bias), ptr(ws), _stream(), _prec()),
              "sparch_gemm_spike16_nt_wp")
    elif spike_scale is not None and USE_SPIKE_GEMM and a16 is not None and USE_SPIKE16:
        tok = timer.start(f"gemm_spike_nt[{M}x{N}x{K}]")
        check(lib.sparch_gemm_spike16_nt(M, N, K, ptr(a16), a16.stride(0), float(spike_scale), ptr(B), B.stride(0),
                                         ptr(C), N, ptr(bias), ptr(ws), _stream(), _prec()), "sparch_gemm_spike16_nt")
    elif spike_scale is not None and USE_SPIKE_GEMM:
        tok = timer.start(f"gemm_spike_nt[{M}x{N}x{K}]")
        check(lib.sparch_gemm_spike_nt(M, N, K, ptr(A), A.stride(0), float(spike_scale), ptr(B), B.stride(0),
                                       ptr(C), N, ptr(bias), ptr(ws), _stream(), _prec()), "sparch_gemm_spike_nt")
    elif a_exact_flag is not None and a_plane is not None and USE_SPIKE_GEMM and DENSE_GEMM == "split6":
        tok = timer.start(f"gemm_auto_nt[{M}x{N}x{K}]")  # a_plane: plane_bf16_exact(A)[0], read when the flag is 1
        check(lib.sparch_gemm_auto16_nt(M, N, K, ptr(A), A.stride(0) or K, ptr(a_plane), a_plane.stride(0), ptr(B),
                                        B.stride(0), ptr(C), N, ptr(bias), ptr(ws), ptr(a_exact_flag), _stream(), _prec()),
              "sparch_gemm_auto16_nt")
    elif a_exact_flag is not None and USE_SPIKE_GEMM and DENSE_GEMM == "split6":
        tok = timer.start(f"gemm_auto_nt[{M}x{N}x{K}]")
        check(lib.sparch_gemm_auto_nt(M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), N, ptr(bias),
                                      ptr(ws), ptr(a_exact_flag), _stream(), _prec()), "sparch_gemm_auto_nt")
    else:
        split6 = DENSE_GEMM == "split6"  # (the fp32-input MFMA kernels of gemm.hip have one precision)
        fn = lib.sparch_gemm6_nt if split6 else lib.sparch_gemm_nt
        tok = timer.start(f"gemm_nt[{M}x{N}x{K}]")
        check(fn(M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), N, ptr(bias), ptr(ws), _stream(),
                 *((_prec(),) if split6 else ())), "sparch_gemm_nt")
    timer.stop(tok)
    return C, ws


def gemm_nn(A, B, b_planes=None):
    """A (M,K) @ B (K,N) -> (M,N).  b_planes: split_planes(B)."""
    M, K = A.shape
    N = B.shape[1]
    C = torch.empty(M, N, dtype=torch.float32, device=A.device)
    tok = timer.start(f"gemm_nn[{M}x{N}x{K}]")
    if b_planes is not None and DENSE_GEMM == "split6":
        check(lib.sparch_gemm6_nn_wp(M, N, K, ptr(A), A.stride(0), ptr(B), ptr(b_planes), B.stride(0), ptr(C), N,
                                     _stream(), _prec()), "sparch_gemm6_nn_wp")
    else:
        split6 = DENSE_GEMM == "split6"
        fn = lib.sparch_gemm6_nn if split6 else lib.sparch_gemm_nn
        check(fn(M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), N, _stream(), *((_prec(),) if split6 else ())),
              "sparch_gemm_nn")
    timer.stop(tok)
    return C


def gemm_tn(A, B, zero_diag=False, spike_side=None, spike_scale=1.0, out=None, b_exact_flag=None, spike16=False,
            b_plane=None):
    """A (K,M)^T @ B (K,N) -> (M,N); contraction over the long leading axis.
    spike_side 0/1: A / B is a spike tensor (entries 0 or spike_scale) -> exact bf16-split MFMA path;
    spike16: that operand is given as a bf16 0/1 plane (torch.bfloat16).
    out: accumulate into this (M,N) tensor instead of allocating."""
    K, M = A.shape
    N = B.shape[1]
    accumulate = out is not None
    C = out if accumulate else torch.empty(M, N, dtype=torch.float32, device=A.device)
    if spike16 and not (spike_side is not None and USE_SPIKE_GEMM):
        raise RuntimeError("internal: a bf16 spike plane needs the spike GEMM path")
    if spike_side is not None and USE_SPIKE_GEMM and spike16:
        nbytes = lib.sparch_gemm_spike_tn_workspace_bytes(M, N, K, _prec())
        ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=A.device)
        tok = timer.start(f"gemm_spike_tn[{M}x{N}x{K}]")
        check(lib.sparch_gemm_spike16_tn(M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), int(spike_side),
                                         float(spike_scale), ptr(C), C.stride(0), int(zero_diag), int(accumulate),
                                         ptr(ws), nbytes, _stream(), _prec()), "sparch_gemm_spike16_tn")
    elif spike_side is not None and USE_SPIKE_GEMM:
        nbytes = lib.sparch_gemm_spike_tn_workspace_bytes(M, N, K, _prec())
        ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=A.device)
        tok = timer.start(f"gemm_spike_tn[{M}x{N}x{K}]")
        check(lib.sparch_gemm_spike_tn(M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), int(spike_side),
                                       float(spike_scale), ptr(C), C.stride(0), int(zero_diag), int(accumulate),
                                       ptr(ws), nbytes, _stream(), _prec()), "sparch_gemm_spike_tn")
    elif b_exact_flag is not None and b_plane is not None and USE_SPIKE_GEMM and DENSE_GEMM == "split6":
        nbytes = lib.sparch_gemm_spike_tn_workspace_bytes(M, (N + 7) // 8 * 8, K, _prec())  # slabs at the plane's padded width
        ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=A.device)
        tok = timer.start(f"gemm_auto_tn[{M}x{N}x{K}]")
        check(lib.sparch_gemm_auto16_tn(M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0) or N, ptr(b_plane),
                                        b_plane.stride(0), ptr(C), C.stride(0), int(zero_diag), int(accumulate),
                                        ptr(b_exact_flag), ptr(ws), nbytes, _stream(), _prec()), "sparch_gemm_auto16_tn")
    elif b_exact_flag is not None and USE_SPIKE_GEMM and DENSE_GEMM == "split6":
        nbytes = lib.sparch_gemm_spike_tn_workspace_bytes(M, N, K, _prec())
        ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=A.device)
        tok = timer.start(f"gemm_auto_tn[{M}x{N}x{K}]")
        check(lib.sparch_gemm_auto_tn(M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), C.stride(0),
                                      int(zero_diag), int(accumulate), ptr(b_exact_flag), ptr(ws), nbytes,
                                      _stream(), _prec()), "sparch_gemm_auto_tn")
    else:
        if spike_side is not None and spike_scale != 1.0:
            raise RuntimeError("internal: fp32 gemm_tn fallback expects unscaled operands")
        split6 = DENSE_GEMM == "split6"
        nbytes = (lib.sparch_gemm_spike_tn_workspace_bytes(M, N, K, _prec()) if split6
                  else lib.sparch_gemm_tn_workspace_bytes(M, N, K))
        ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=A.device)
        fn = lib.sparch_gemm6_tn if split6 else lib.sparch_gemm_tn
        tok = timer.start(f"gemm_tn[{M}x{N}x{K}]")
        check(fn(M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), C.stride(0), int(zero_diag),
                 int(accumulate), ptr(ws), nbytes, _stream(), *((_prec(),) if split6 else ())), "sparch_gemm_tn")
    timer.stop(tok)
    return C


def _colsum(x2d):
    M, H = x2d.shape
    out = torch.empty(H, dtype=torch.float32, device=x2d.device)
    nbytes = lib.sparch_bn_bwd_workspace_bytes(M, H)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x2d.device)
    check(lib.sparch_colsum(M, H, ptr(x2d), ptr(out), ptr(ws), nbytes, _stream()), "sparch_colsum")
    return out


def _finish_param_grads(ws, rows, H, raws, lims):
    """ws (n,rows,H) partials -> list of (H,) grads gated by the clamp range (torch.clamp backward)."""
    import ctypes

    n = len(raws)
    outs = [torch.empty(H, dtype=torch.float32, device=ws.device) for _ in range(n)]
    raw_arr = (ctypes.c_void_p * n)(*[(r.data_ptr() if r is not None else None) for r in raws])  # None: no clamp gate
    out_arr = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    lim_arr = (ctypes.c_float * (2 * n))(*[v for lo_hi in lims for v in lo_hi])
    check(lib.sparch_colsum_clamped(n, rows, H, ptr(ws), raw_arr, lim_arr, out_arr, _stream()),
          "sparch_colsum_clamped")
    return outs


class _Norm:
    """Forward/backward of the normalisation on the (M,H) projection.  BatchNorm is folded into
    (scale, shift) consumed by the cell kernel; LayerNorm materialises the normalised tensor."""

    @staticmethod
    def forward(mode, Wx_raw, colstat, weight, bias, running_mean, running_var, training, dup, nbt=None, ln_width=None):
        """nbt: BatchNorm's num_batches_tracked (int64 device tensor) — incremented by the finalize kernel itself,
        under the status-word guard, instead of by a separate host-side `+= 1` (None: the caller keeps doing that).
        ln_width: LayerNorm's width when the layer runs zero-padded to more columns (None: all H columns)."""
        M, H = Wx_raw.shape
        dev = Wx_raw.device
        if mode == "batchnorm":
            scale = torch.empty(H, dtype=torch.float32, device=dev)
            shift = torch.empty(H, dtype=torch.float32, device=dev)
            mean = torch.empty(H, dtype=torch.float32, device=dev)
            invstd = torch.empty(H, dtype=torch.float32, device=dev)
            n_tiles, rows = (M + 127) // 128, M
            if training and SYNC_BN is not None and SYNC_BN["world"] > 1:
                import torch.distributed as dist

                world = SYNC_BN["world"]
                parts = [torch.empty_like(colstat) for _ in range(world)]
                dist.all_gather(parts, colstat, group=SYNC_BN["group"])
                colstat = torch.stack([q.view(2, n_tiles, H) for q in parts], dim=1).reshape(-1)  # (2, world*tiles, H)
                n_tiles, rows = n_tiles * world, M * world
            check(lib.sparch_bn_finalize(H, rows, n_tiles, dup, ptr(colstat), ptr(weight), ptr(bias),
                                         ptr(running_mean), ptr(running_var), BN_MOMENTUM, NORM_EPS,
                                         int(training), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                                         ptr(status_word(dev)), ptr(nbt), _stream()), "sparch_bn_finalize")
            return Wx_raw, scale, shift, (mean, invstd)
        if mode == "layernorm":
            y = torch.empty_like(Wx_raw)
            mu = torch.empty(M, dtype=torch.float32, device=dev)
            rstd = torch.empty(M, dtype=torch.float32, device=dev)
            check(lib.sparch_layernorm_fwd(M, H, ln_width or H, ptr(Wx_raw), ptr(weight), ptr(bias), NORM_EPS, ptr(y),
                                           ptr(mu), ptr(rstd), _stream()), "sparch_layernorm_fwd")
            return y, None, None, (mu, rstd)
        return Wx_raw, None, None, None

    @staticmethod
    def backward(mode, dy, Wx_raw, weight, saved, training, sums=None, planes=False, dy2=None, keep_fp32=True,
                 ln_width=None):
        """dy (M,H) grad wrt the normalised projection -> (dx_raw, dweight, dbias). May overwrite dy.
        sums = (dbeta, dgamma) when the cell's backward kernel already produced BatchNorm's column sums.
        planes=True (batchnorm, H % 8 == 0): returns (dx_raw or None, dweight, dbias, dx_planes) with dx_planes the
        (3, M, H) bf16 planes of dx_raw; keep_fp32=False skips the fp32 tensor (dx_raw is None).  dy2: the second
        direction's gradient of a bidirectional layer, added in the same pass (dy + dy2)."""
        M, H = dy.shape
        dev = dy.device
        if mode == "batchnorm":
            mean, invstd = saved  # batch statistics (train) or running statistics (eval)
            if sums is not None:
                dbeta, dgamma = sums
            else:
                dgamma = torch.empty(H, dtype=torch.float32, device=dev)
                dbeta = torch.empty(H, dtype=torch.float32, device=dev)
                nbytes = lib.sparch_bn_bwd_workspace_bytes(M, H)
                ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
                check(lib.sparch_bn_bwd_reduce(M, H, ptr(dy), ptr(Wx_raw), ptr(mean), ptr(invstd), ptr(dgamma),
                                               ptr(dbeta), ptr(ws), nbytes, _stream()), "sparch_bn_bwd_reduce")
            if training and SYNC_BN is not None and SYNC_BN["world"] > 1:
                import torch.distributed as dist

                red = torch.stack([dgamma, dbeta])
                dist.all_reduce(red, op=dist.ReduceOp.SUM, group=SYNC_BN["group"])
                red.mul_(1.0 / SYNC_BN["world"])  # the apply kernel divides by the LOCAL row count
                cg, cb = red[0], red[1]
            elif training:
                cg, cb = dgamma, dbeta
            else:  # fixed statistics: dx = dy * gamma * invstd, i.e. the batch-coupling terms vanish
                cg = torch.zeros(H, dtype=torch.float32, device=dev)
                cb = cg
            if planes:
                dxp = torch.empty(3, M, H, dtype=torch.bfloat16, device=dev)
                dx_out = dy if keep_fp32 else None
                tok = timer.start(f"bn_bwd_apply_planes[{M}x{H}]")
                check(lib.sparch_bn_bwd_apply_planes(M, H, ptr(dy), ptr(dy2), ptr(Wx_raw), ptr(mean), ptr(invstd),
                                                     ptr(weight), ptr(cg), ptr(cb), ptr(dxp), ptr(dx_out), _stream()),
                      "sparch_bn_bwd_apply_planes")
                timer.stop(tok)
                return dx_out, dgamma, dbeta, dxp
            check(lib.sparch_bn_bwd_apply(M, H, ptr(dy), ptr(Wx_raw), ptr(mean), ptr(invstd), ptr(weight),
                                          ptr(cg), ptr(cb), ptr(dy), _stream()), "sparch_bn_bwd_apply")
            return dy, dgamma, dbeta
        if mode == "layernorm":
            mu, rstd = saved
            dgamma = torch.empty(H, dtype=torch.float32, device=dev)
            dbeta = torch.empty(H, dtype=torch.float32, device=dev)
            dx = torch.empty_like(dy)
            nbytes = lib.sparch_bn_bwd_workspace_bytes(M, H)
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            check(lib.sparch_layernorm_bwd(M, H, ln_width or H, ptr(dy), ptr(Wx_raw), ptr(mu), ptr(rstd), ptr(weight),
                                           ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, _stream()),
                  "sparch_layernorm_bwd")
            return dx, dgamma, dbeta
        return dy, None, None


# ----------------------------------------------------------------------------- cells (given Wx)
_placeholder_zero = {}  # one element per device (a fresh torch.zeros(1) per layer and step was a fill kernel each)


def spike_placeholder(B, T, F, device):
    """(B,T,F) fp32 stand-in for a spike tensor whose only consumer reads its bf16 plane: one element of storage,
    expanded — autograd needs the edge's shape and dtype, nobody reads the values."""
    key = str(device)
    z = _placeholder_zero.get(key)
    if z is None:
        z = _placeholder_zero[key] = torch.zeros(1, dtype=torch.float32, device=device)
    return z.expand(B, T, F)


def cell_forward(kind, Wx, scale, shift, p, u0, w0, s0, *, B, dirs, theta, p_drop, seed, steps_per_launch=None,
                 want_s_out=True, want_saves=True):
    """Run one spiking cell over the whole sequence on the device.

    Wx (B,T,H) raw projection (+ optional per-column scale/shift); u0/w0/s0 (B*dirs,H).
    Returns (s_out (B,T,H*dirs), count (H*dirs) int32, saved, s16) where saved feeds cell_backward and s16 is
    s_out != 0 as a bf16 plane (or None).  want_s_out=False (round 3): the fp32 tensor is not written — s_out is
    None — when the bf16 plane exists, i.e. for a layer whose output only feeds the next layer's spike GEMMs
    (the reference materialises it because its next op is a dense nn.Linear, snns.py:261; here it was 4 of the
    14 bytes a recurrent forward step stores per element).  want_saves=False (LIF / adLIF, nothing will be
    differentiated — validation and test forwards, exp.py:405-518): u / w are not saved either (8 of an adLIF step's
    14 bytes); `saved` is then (None, None)."""
    _, T, H = Wx.shape
    Bp = B * dirs
    dev = Wx.device
    k = KIND[kind]
    adaptive, recurrent = bool(k & 1), bool(k & 2)
    # the same spikes as a bf16 0/1 plane for the GEMMs of the next layer (rows must stay 16-byte aligned)
    s16 = torch.empty(B, T, H * dirs, dtype=torch.bfloat16, device=dev) if (USE_SPIKE_GEMM and USE_SPIKE16) else None
    want_s_out = want_s_out or s16 is None
    s_out = torch.empty(B, T, H * dirs, dtype=torch.float32, device=dev) if want_s_out else None
    L = steps_per_launch if steps_per_launch is not None else rec_steps_per_launch(T)
    # bf16 saves: the recurrent kernels take them for whole-sequence launches only (a chunked forward resumes
    # from the saved state, which must then be exact)
    save16 = SAVE_BF16 and H % 4 == 0 and (not recurrent or (L >= T and not rec_step_path(H)))
    sdt = torch.bfloat16 if save16 else torch.float32
    want_saves = want_saves or recurrent  # (the recurrent kernels always save)
    u_save = torch.empty(Bp, T, H, dtype=sdt, device=dev) if want_saves else None
    w_save = torch.empty(Bp, T, H, dtype=sdt, device=dev) if (adaptive and want_saves) else None
    count = torch.zeros(H * dirs, dtype=torch.int32, device=dev)
    if not recurrent:
        tok = timer.start(f"cell_fwd[{kind}]")
        check(lib.sparch_cell_fwd(k, B, dirs, T, H, ptr(Wx), ptr(scale), ptr(shift), ptr(p["alpha"]),
                                  ptr(p.get("beta")), ptr(p.get("a")), ptr(p.get("b")), ptr(u0), ptr(w0),
                                  ptr(s0), theta, p_drop, seed, ptr(s_out), ptr(s16), ptr(u_save), ptr(w_save),
                                  int(save16), ptr(count), _stream()), "sparch_cell_fwd")
        timer.stop(tok)
    else:
        if H % 4 != 0:
            # The recurrent kernels own 4 columns per thread.  Any other width (the reference takes any nb_hiddens,
            # snns.py:608-661) runs zero-padded to the next multiple of 4: a padded neuron has no input, no
            # recurrent weights and u0 = 0, so it never spikes and feeds nothing; outputs are sliced back.
            H4 = (H + 3) // 4 * 4
            padh = lambda t_: None if t_ is None else torch.nn.functional.pad(t_, (0, H4 - H))  # noqa: E731
            pp = {k_: (torch.nn.functional.pad(v, (0, H4 - H, 0, H4 - H)) if k_ == "V" else padh(v)) for k_, v in p.items()}
            s_p, count_p, saved_p, s16_p = cell_forward(kind, padh(Wx), padh(scale), padh(shift), pp, padh(u0), padh(w0),
                                                        padh(s0), B=B, dirs=dirs, theta=theta, p_drop=p_drop, seed=seed,
                                                        steps_per_launch=steps_per_launch, want_s_out=want_s_out)
            cut = lambda t_: None if t_ is None else t_.view(B, T, dirs, H4)[..., :H].reshape(B, T, dirs * H).contiguous()  # noqa: E731
            return (cut(s_p), count_p.view(dirs, H4)[:, :H].reshape(-1).contiguous(), saved_p, cut(s16_p))
        V = p["V"]
        if rec_step_path(H):
            # One launch per time step, the recurrent product s_{t-1} @ V between the steps on the exact
            # spike GEMM (3 bf16 planes of V, fp32 accumulate): same arithmetic as the persistent kernel.
            vmask = torch.empty(H, H, dtype=torch.float32, device=dev)
            check(lib.sparch_vmask(H, ptr(V), ptr(vmask), _stream()), "sparch_vmask")
            vmask_t = vmask.t().contiguous()              # (H_out, H_in): the NT operand
            rec = gemm_nn(s0, vmask)                      # t = 0: s0 is uniform noise, not binary
            s_step = torch.empty(Bp, H, dtype=torch.bfloat16, device=dev)
            tok = timer.start(f"rec_cell_fwd_steps[{kind}]")
            for t in range(T):
                if t > 0:
                    rec, _ = gemm_nt(s_step, vmask_t, spike_scale=1.0, a16=s_step)
                check(lib.sparch_rec_cell_step_fwd(k, B, dirs, T, H, t, ptr(Wx), ptr(scale), ptr(shift),
                                                   ptr(p["alpha"]), ptr(p.get("beta")), ptr(p.get("a")),
                                                   ptr(p.get("b")), ptr(rec), ptr(u0), ptr(w0), ptr(s0), theta,
                                                   p_drop, seed, ptr(s_out), ptr(s16), ptr(u_save), ptr(w_save),
                                                   ptr(count), ptr(s_step), _stream()), "sparch_rec_cell_step_fwd")
            timer.stop(tok)
            return s_out, count, (u_save, w_save), s16
        # forward fragments, backward fragments (kept for cell_backward) and the masked copy: one launch
        vpack = torch.empty(lib.sparch_vpack_bytes(H) // 4, dtype=torch.float32, device=dev)
        vpack_t = torch.empty_like(vpack)
        vmask = torch.empty(H, H, dtype=torch.float32, device=dev)
        check(lib.sparch_vpack_both(H, ptr(V), ptr(vpack), ptr(vpack_t), ptr(vmask), _stream(), _prec()), "sparch_vpack_both")
        # t = 0 drive: s0 is uniform noise, not binary (snns.py:559/702): a (B', H, H) dense product, split-K so
        # that its 16 output tiles become a full grid (38 -> ~15 us at B' = 256, H = 1024)
        rec0 = _gemm_small(s0, vmask, nn=True)
        nbytes = lib.sparch_rec_chan_bytes(Bp, T, H)
        chan = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)  # (ceil: the byte count need not be a multiple of 8)
        tok = timer.start(f"rec_cell_fwd[{kind}]")
        with _persistent_launch():
            check(lib.sparch_rec_cell_fwd(k, B, dirs, T, H, ptr(Wx), ptr(scale), ptr(shift), ptr(p["alpha"]),
                                          ptr(p.get("beta")), ptr(p.get("a")), ptr(p.get("b")), ptr(vpack),
                                          ptr(rec0), ptr(u0), ptr(w0), ptr(s0), theta, p_drop, seed, ptr(s_out),
                                          ptr(s16), ptr(u_save), ptr(w_save), int(save16), ptr(count), ptr(chan),
                                          nbytes, ptr(status_word(dev)), L, _stream(), _prec()), "sparch_rec_cell_fwd")
        timer.stop(tok)
        return s_out, count, (u_save, w_save, vpack_t), s16
    return s_out, count, (u_save, w_save), s16


def cell_backward(kind, g_out, g_rate, p, u0, w0, s0, saved, *, B, dirs, T, H, theta, p_drop, seed,
                  steps_per_launch=None, bn=None):
    """Reverse-time pass.  Returns dWx (B*dirs,T,H) [virtual rows, original time index],
    param grads dict (alpha[,beta,a,b][,V]) — plus, with bn = (Wx_raw (B,T,H), mean, invstd), the entry
    "bn_sums" = (dbeta, dgamma): BatchNorm backward's column sums, accumulated by the same kernel."""
    Bp = B * dirs
    dev = g_out.device
    k = KIND[kind]
    adaptive, recurrent = bool(k & 1), bool(k & 2)
    u_save, w_save = saved[0], saved[1]
    if recurrent and u_save.shape[-1] != H:  # forward ran zero-padded to a multiple of 4 columns (see cell_forward)
        H4 = u_save.shape[-1]
        padh = lambda t_: None if t_ is None else torch.nn.functional.pad(t_, (0, H4 - H))  # noqa: E731
        pp = {k_: (torch.nn.functional.pad(v, (0, H4 - H, 0, H4 - H)) if k_ == "V" else padh(v)) for k_, v in p.items()}
        g_p = padh(g_out.reshape(B, T, dirs, H)).reshape(B, T, dirs * H4)
        gr_p = None if g_rate is None else padh(g_rate.reshape(dirs, H)).reshape(dirs * H4)
        dWx_p, gp = cell_backward(kind, g_p, gr_p, pp, padh(u0), padh(w0), padh(s0), saved, B=B, dirs=dirs, T=T, H=H4,
                                  theta=theta, p_drop=p_drop, seed=seed, steps_per_launch=steps_per_launch, bn=None)
        grads = {k_: (v[:H, :H].contiguous() if k_ == "V" else v[:H].contiguous()) for k_, v in gp.items()}
        return dWx_p[..., :H].contiguous(), grads
    vpack_fwd_made = saved[2] if len(saved) > 2 else None  # backward fragments of V packed by cell_forward
    save16 = u_save.dtype == torch.bfloat16
    dWx = torch.empty(Bp, T, H, dtype=torch.float32, device=dev)
    n_base = 6 if recurrent else 4
    ws = torch.empty(n_base + (2 if bn is not None else 0), Bp, H, dtype=torch.float32, device=dev)
    bn_x, bn_mean, bn_invstd = bn if bn is not None else (None, None, None)
    grads = {}
    if not recurrent:
        tok = timer.start(f"cell_bwd[{kind}]")
        check(lib.sparch_cell_bwd(k, B, dirs, T, H, ptr(g_out), ptr(g_rate), ptr(u_save), ptr(w_save), int(save16),
                                  ptr(p["alpha"]), ptr(p.get("beta")), ptr(p.get("a")), ptr(p.get("b")), ptr(u0),
                                  ptr(w0), ptr(s0), theta, p_drop, seed, ptr(dWx), ptr(ws), ptr(bn_x), ptr(bn_mean),
                                  ptr(bn_invstd), _stream()), "sparch_cell_bwd")
        timer.stop(tok)
    else:
        V = p["V"]
        s_prev = torch.empty(Bp, T, H, dtype=torch.bfloat16, device=dev)  # bf16 0/1 plane
        if rec_step_path(H):
            # reverse-time steps with dWx_{t+1} @ V^T between them (exact six-term split GEMM)
            vmask = torch.empty(H, H, dtype=torch.float32, device=dev)
            check(lib.sparch_vmask(H, ptr(V), ptr(vmask), _stream()), "sparch_vmask")
            dwx_step = torch.empty(Bp, H, dtype=torch.float32, device=dev)
            rec = None
            tok = timer.start(f"rec_cell_bwd_steps[{kind}]")
            for t in range(T - 1, -1, -1):
                if t + 1 < T:
                    rec, _ = gemm_nt(dwx_step, vmask)     # rec[b,i] = sum_j dWx[b,j] Vm[i,j]
                check(lib.sparch_rec_cell_step_bwd(k, B, dirs, T, H, t, ptr(g_out), ptr(g_rate), ptr(u_save),
                                                   ptr(w_save), ptr(p["alpha"]), ptr(p.get("beta")),
                                                   ptr(p.get("a")), ptr(p.get("b")), ptr(rec), ptr(u0), ptr(w0),
                                                   ptr(s0), theta, p_drop, seed, ptr(dWx), ptr(s_prev), ptr(ws),
                                                   ptr(bn_x), ptr(bn_mean), ptr(bn_invstd), ptr(dwx_step),
                                                   _stream()), "sparch_rec_cell_step_bwd")
            timer.stop(tok)
        else:
            if vpack_fwd_made is not None:
                vpack_t = vpack_fwd_made
            else:
                vpack_t = torch.empty(lib.sparch_vpack_bytes(H) // 4, dtype=torch.float32, device=dev)
                check(lib.sparch_vpack(H, ptr(V), 1, ptr(vpack_t), None, _stream(), _prec()), "sparch_vpack")
            nbytes = lib.sparch_rec_chan_bytes(Bp, T, H)
            chan = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)  # (ceil: the byte count need not be a multiple of 8)
            L = steps_per_launch if steps_per_launch is not None else rec_steps_per_launch(T)
            tok = timer.start(f"rec_cell_bwd[{kind}]")
            with _persistent_launch():
                check(lib.sparch_rec_cell_bwd(k, B, dirs, T, H, ptr(g_out), ptr(g_rate), ptr(u_save), ptr(w_save),
                                              int(save16), ptr(p["alpha"]), ptr(p.get("beta")), ptr(p.get("a")), ptr(p.get("b")),
                                              ptr(vpack_t), ptr(u0), ptr(w0), ptr(s0), theta, p_drop, seed, ptr(dWx),
                                              ptr(s_prev), ptr(ws), ptr(bn_x), ptr(bn_mean), ptr(bn_invstd), ptr(chan),
                                              nbytes, ptr(status_word(dev)), L, _stream(), _prec()), "sparch_rec_cell_bwd")
            timer.stop(tok)
        # dV = sum_t s_{t-1}^T (1-alpha) du_t with the diagonal zeroed (mask at snns.py:566/712):
        # binary rows t >= 1 on the exact bf16-split path, plus the t = 0 term with the non-binary s0
        # (cell step 0 sits at original time 0 for the forward direction, T-1 for the flipped one)
        if USE_SPIKE_GEMM:
            dV = gemm_tn(s_prev.view(Bp * T, H), dWx.view(Bp * T, H), zero_diag=True, spike_side=0, spike16=True)
        else:  # fp32-MFMA comparison path (tests)
            dV = gemm_tn(s_prev.view(Bp * T, H).float(), dWx.view(Bp * T, H), zero_diag=True)
        for dd in range(dirs):  # s_prev rows of cell step 0 are zero on both paths: add the s0 term
            rows = slice(dd * B, (dd + 1) * B)
            gemm_tn(s0[rows], dWx[rows, (T - 1) if dd else 0, :], zero_diag=True, out=dV)
        grads["V"] = dV
    names = ["alpha"] + (["beta", "a", "b"] if adaptive else [])
    lims = [ALPHA_LIM] + ([BETA_LIM, A_LIM, B_LIM] if adaptive else [])
    if bn is None:
        outs = _finish_param_grads(ws, Bp, H, [p[n] for n in names], lims)
    else:
        # one launch for the parameter planes and the two BatchNorm planes behind them (no clamp gate there; the
        # planes in between, if any, are summed too and dropped)
        n_all = n_base + 2
        outs = _finish_param_grads(ws, Bp, H, [p[n] for n in names] + [None] * (n_all - len(names)),
                                   lims + [(0.0, 0.0)] * (n_all - len(names)))
        grads["bn_sums"] = (outs[n_base], outs[n_base + 1])
    grads.update(dict(zip(names, outs)))
    return dWx, grads


# ----------------------------------------------------------------------------- layer Functions
class SpikingLayerFn(torch.autograd.Function):
    """x (B,T,K) -> (s (B,T,H*dirs), firing_rate (H*dirs)) for LIF / adLIF / RLIF / RadLIF."""

    @staticmethod
    def forward(ctx, cfg, x, W, Wb, nw, nb, alpha, beta, a, b, V, u0, w0, s0):
        _require_device(x, "input")
        _require_device(W, "layer parameters")
        kind, norm, dirs = cfg["kind"], cfg["normalization"], cfg["dirs"]
        training, theta, p_drop, seed = cfg["training"], cfg["theta"], cfg["p_drop"], cfg["seed"]
        in_scale = cfg.get("in_spike_scale")  # input is a spike train of ours: entries 0 or in_scale
        plane_in = ((in_scale is not None and cfg.get("in_spike16") is not None and USE_SPIKE_GEMM and USE_SPIKE16)
                    or cfg.get("in_plane") is not None)
        if not plane_in:  # (an input read through its bf16 plane may be a placeholder: never touch its values)
            x = _f32c(x)
        B, T, K = x.shape
        H = W.shape[0]
        M = B * T
        x2 = x.view(M, K)
        use_bn_stats = norm == "batchnorm" and training
        # otherwise (network input): let the device decide whether x is bf16-exact (binned spike counts are)
        xplane = None
        if cfg.get("in_plane") is not None:
            xplane, xflag = cfg["in_plane"]  # the input came as bytes (input_from_counts): plane made, flag = 1
        elif in_scale is None and USE_SPIKE_GEMM and USE_SPIKE16 and USE_INPUT_PLANE and DENSE_GEMM == "split6":
            xplane, xflag = plane_bf16_exact(x2)  # one pass: the flag AND the plane the two GEMMs read when it is 1
        else:
            xflag = flag_bf16_exact(x2) if (in_scale is None and USE_SPIKE_GEMM) else None
        ctx.xflag, ctx.xplane = xflag, xplane
        x16 = cfg.get("in_spike16") if in_scale is not None else None
        x16 = x16.view(M, K) if x16 is not None else None
        # the weights' bf16 planes, split once here for backward's dx GEMM (775 -> 732 us).  The projection itself
        # converts W on the fly by default: every workgroup streams all of W, and 4 MB of fp32 stay resident in
        # the XCD's 4 MB L2 where 6 MB of planes do not (0.42 against 0.61 GB of L2 misses per launch, 1.8 % faster;
        # SPARCH_PRESPLIT_NT=1 hands it the planes too)
        need_dx = ctx.needs_input_grad[1]
        planes_ok = K % 32 == 0 and H >= 128
        w_planes = split_planes(W) if planes_ok and (need_dx or (PRESPLIT_NT and x16 is not None)) else None
        ctx.w_planes = w_planes if need_dx else None
        Wx_raw, colstat = gemm_nt(x2, W, Wb, colstat=use_bn_stats, spike_scale=in_scale, a_exact_flag=xflag, a16=x16,
                                  b_planes=w_planes if PRESPLIT_NT else None, a_plane=xplane)  # snns.py:261
        Wx_in, scale, shift, nsaved = _Norm.forward(norm, Wx_raw, colstat, nw, nb, cfg.get("running_mean"),
                                                    cfg.get("running_var"), training, dirs,
                                                    nbt=cfg.get("num_batches_tracked"))  # 264-266
        p = {"alpha": alpha, "beta": beta, "a": a, "b": b, "V": V}
        p = {k_: v for k_, v in p.items() if v is not None}
        if cfg.get("states_ready") is not None:  # initial states uploaded on a side stream (snns._rand_batch)
            cfg["states_ready"]()
        s_out, count, saved, s16 = cell_forward(kind, Wx_in.view(B, T, H), scale, shift, p, u0, w0, s0, B=B,
                                                dirs=dirs, theta=theta, p_drop=p_drop, seed=seed,
                                                want_s_out=cfg.get("fp32_out", True),
                                                want_saves=any(ctx.needs_input_grad))
        if s_out is None:
            s_out = spike_placeholder(B, T, H * dirs, x.device)
        inv_keep = 1.0 / (1.0 - p_drop)
        rate = count * (inv_keep / float(B * T))  # snns.py:174 on post-dropout spikes (int32 * float -> fp32, one kernel)
        ctx.cfg = cfg
        ctx.shape = (B, T, K, H)
        ctx.nsaved = nsaved
        ctx.cell_saved = saved
        ctx.set_materialize_grads(False)  # no zero tensors for unused outputs (s16 is as large as s in bf16)
        ctx.save_for_backward(x2, W, nw, alpha, beta, a, b, V, u0, w0, s0,
                              Wx_raw if norm in ("batchnorm", "layernorm") else None)
        if s16 is None:  # keep the output arity fixed
            s16 = torch.empty(0, dtype=torch.bfloat16, device=x.device)
        ctx.mark_non_differentiable(s16)
        return s_out, rate, s16

    @staticmethod
    def backward(ctx, g_s, g_rate, _g_s16=None):
        cfg = ctx.cfg
        kind, norm, dirs = cfg["kind"], cfg["normalization"], cfg["dirs"]
        B, T, K, H = ctx.shape
        x2, W, nw, alpha, beta, a, b, V, u0, w0, s0, Wx_raw = ctx.saved_tensors
        M = B * T
        dev = x2.device
        if g_s is None:
            g_s = torch.zeros(B, T, H * dirs, dtype=torch.float32, device=dev)
        g_s = _f32c(g_s)
        if g_rate is not None:
            g_rate = _f32c(g_rate)
        p = {"alpha": alpha, "beta": beta, "a": a, "b": b, "V": V}
        p = {k_: v for k_, v in p.items() if v is not None}
        bn = (Wx_raw, ctx.nsaved[0], ctx.nsaved[1]) if (norm == "batchnorm" and FUSE_BN_SUMS and H % 4 == 0) else None
        dWx, pg = cell_backward(kind, g_s, g_rate, p, u0, w0, s0, ctx.cell_saved, B=B, dirs=dirs, T=T, H=H,
                                theta=cfg["theta"], p_drop=cfg["p_drop"], seed=cfg["seed"], bn=bn)
        ctx.cell_saved = None
        in_scale = cfg.get("in_spike_scale")
        x16 = cfg.get("in_spike16") if (in_scale is not None and USE_SPIKE16) else None
        # dx as bf16 planes (made once by the BatchNorm pass) for the dW and dX products: a hidden layer fed by a spike
        # plane, exact mode, shapes the pipelined plane kernels take (whole tiles, 32-deep K tiles)
        use_planes = ((USE_DX_PLANES is True or (USE_DX_PLANES == "auto" and dirs == 2))
                      and norm == "batchnorm" and x16 is not None and USE_SPIKE_GEMM and DENSE_GEMM == "split6"
                      and _precision == 0 and H % 32 == 0 and K % 32 == 0 and H >= 256 and K >= 256 and M >= 256
                      and M % 32 == 0 and (not ctx.needs_input_grad[1] or ctx.w_planes is not None))
        need_bias = ctx.needs_input_grad[3]
        if use_planes:
            dy2 = dWx[B:].view(M, H) if dirs == 2 else None  # second direction: added by the same pass
            dx_raw, dnw, dnb, dxp = _Norm.backward(norm, dWx[:B].view(M, H), Wx_raw, nw, ctx.nsaved, cfg["training"],
                                                   sums=pg.get("bn_sums"), planes=True, dy2=dy2, keep_fp32=need_bias)
            nbytes = lib.sparch_gemm_spike_tn_workspace_bytes(H, K, M, _prec())
            ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=dev)
            dW = torch.empty(H, K, dtype=torch.float32, device=dev)
            x16m = x16.view(M, K)
            tok = timer.start(f"gemm_spike_tn[{H}x{K}x{M}]")
            check(lib.sparch_gemm_spike16_tn_ap(H, K, M, ptr(dx_raw), ptr(dxp), H, ptr(x16m), x16m.stride(0), float(in_scale),
                                                ptr(dW), K, 0, 0, ptr(ws), nbytes, _stream(), _prec()),
                  "sparch_gemm_spike16_tn_ap")
            timer.stop(tok)
            dWb = _colsum(dx_raw) if need_bias else None
            dx = None
            if ctx.needs_input_grad[1]:
                dx = torch.empty(M, K, dtype=torch.float32, device=dev)
                tok = timer.start(f"gemm_nn[{M}x{K}x{H}]")
                check(lib.sparch_gemm6_nn_pp(M, K, H, ptr(dx_raw), ptr(dxp), H, ptr(W), ptr(ctx.w_planes), K, ptr(dx), K,
                                             _stream(), _prec()), "sparch_gemm6_nn_pp")
                timer.stop(tok)
                dx = dx.view(B, T, K)
        else:
            if dirs == 2:  # both directions share the projection rows (snns.py:252-254)
                dy = torch.empty(B, T, H, dtype=torch.float32, device=dev)
                check(lib.sparch_add_halves(M * H, ptr(dWx), ptr(dy), _stream()), "sparch_add_halves")
            else:
                dy = dWx
            dy = dy.view(M, H)
            dx_raw, dnw, dnb = _Norm.backward(norm, dy, Wx_raw, nw, ctx.nsaved, cfg["training"], sums=pg.get("bn_sums"))
            if x16 is not None and USE_SPIKE_GEMM:
                dW = gemm_tn(dx_raw, x16.view(M, K), spike_side=1, spike_scale=in_scale, spike16=True)
            elif in_scale is not None and USE_SPIKE_GEMM:
                dW = gemm_tn(dx_raw, x2, spike_side=1, spike_scale=in_scale)  # (H,K) = dx_raw^T x, x spikes
            else:
                dW = gemm_tn(dx_raw, x2, b_exact_flag=ctx.xflag, b_plane=ctx.xplane)
            dWb = _colsum(dx_raw) if need_bias else None
            dx = gemm_nn(dx_raw, W, b_planes=ctx.w_planes).view(B, T, K) if ctx.needs_input_grad[1] else None
        ctx.w_planes = ctx.xplane = None
        return (None, dx, dW, dWb, dnw, dnb, pg.get("alpha"), pg.get("beta"), pg.get("a"), pg.get("b"),
                pg.get("V"), None, None, None)


class ReadoutLayerFn(torch.autograd.Function):
    """x (B,T,K) -> out (B,C): softmax-sum readout (snns.py:793-825)."""

    @staticmethod
    def forward(ctx, cfg, x, W, Wb, nw, nb, alpha, u0):
        _require_device(x, "input")
        _require_device(W, "layer parameters")
        norm, training = cfg["normalization"], cfg["training"]
        plane_in = (cfg.get("in_spike_scale") is not None and cfg.get("in_spike16") is not None
                    and USE_SPIKE_GEMM and USE_SPIKE16)
        if not plane_in:
            x = _f32c(x)
        B, T, K = x.shape
        C = W.shape[0]
        if C > 256:
            raise ValueError(f"sparch_amd: the readout kernels handle at most 256 classes (got {C})")
        M = B * T
        x2 = x.view(M, K)
        x16 = cfg.get("in_spike16") if cfg.get("in_spike_scale") is not None else None
        Wx_raw, colstat = gemm_nt(x2, W, Wb, colstat=(norm == "batchnorm" and training),
                                  spike_scale=cfg.get("in_spike_scale"),
                                  a16=x16.view(M, K) if x16 is not None else None)  # snns.py:796
        Wx_in, scale, shift, nsaved = _Norm.forward(norm, Wx_raw, colstat, nw, nb, cfg.get("running_mean"),
                                                    cfg.get("running_var"), training, 1,
                                                    nbt=cfg.get("num_batches_tracked"))  # 799-801
        out = torch.empty(B, C, dtype=torch.float32, device=x.device)
        u_save = torch.empty(B, T, C, dtype=torch.float32, device=x.device)
        check(lib.sparch_readout_fwd(B, T, C, ptr(Wx_in), ptr(scale), ptr(shift), ptr(alpha), ptr(u0), ptr(out),
                                     ptr(u_save), _stream()), "sparch_readout_fwd")
        ctx.cfg = cfg
        ctx.shape = (B, T, K, C)
        ctx.nsaved = nsaved
        ctx.save_for_backward(x2, W, nw, alpha, u0, u_save,
                              Wx_raw if norm in ("batchnorm", "layernorm") else None)
        return out

    @staticmethod
    def backward(ctx, g_out):
        cfg = ctx.cfg
        norm = cfg["normalization"]
        B, T, K, C = ctx.shape
        x2, W, nw, alpha, u0, u_save, Wx_raw = ctx.saved_tensors
        M = B * T
        dev = x2.device
        g_out = _f32c(g_out)
        dWx = torch.empty(B, T, C, dtype=torch.float32, device=dev)
        fuse = norm == "batchnorm" and FUSE_BN_SUMS
        ws = torch.empty(3 if fuse else 1, B, C, dtype=torch.float32, device=dev)
        check(lib.sparch_readout_bwd(B, T, C, ptr(g_out), ptr(Wx_raw) if fuse else None,
                                     ptr(ctx.nsaved[0]) if fuse else None, ptr(ctx.nsaved[1]) if fuse else None,
                                     ptr(u_save), ptr(alpha), ptr(u0), ptr(dWx), ptr(ws), _stream()),
              "sparch_readout_bwd")
        if fuse:  # dalpha and BatchNorm's two column sums: one launch
            dalpha, s1, s2 = _finish_param_grads(ws, B, C, [alpha, None, None], [ALPHA_LIM, (0.0, 0.0), (0.0, 0.0)])
            sums = (s1, s2)
        else:
            (dalpha,) = _finish_param_grads(ws, B, C, [alpha], [ALPHA_LIM])
            sums = None
        dy = dWx.view(M, C)
        dx_raw, dnw, dnb = _Norm.backward(norm, dy, Wx_raw, nw, ctx.nsaved, cfg["training"], sums=sums)
        in_scale = cfg.get("in_spike_scale")
        x16 = cfg.get("in_spike16") if (in_scale is not None and USE_SPIKE16) else None
        if x16 is not None and USE_SPIKE_GEMM:
            dW = gemm_tn(dx_raw, x16.view(M, K), spike_side=1, spike_scale=in_scale, spike16=True)
        elif in_scale is not None and USE_SPIKE_GEMM:
            dW = gemm_tn(dx_raw, x2, spike_side=1, spike_scale=in_scale)
        else:
            dW = gemm_tn(dx_raw, x2)
        dWb = _colsum(dx_raw) if ctx.needs_input_grad[3] else None
        dx = gemm_nn(dx_raw, W).view(B, T, K) if ctx.needs_input_grad[1] else None
        return None, dx, dW, dWb, dnw, dnb, dalpha, None


class SpikingCellFn(torch.autograd.Function):
    """A cell in isolation: Wx (B',T,H) already projected/normalised -> spikes (B',T,H).
    Mirrors the reference's `_lif_cell`/`_adlif_cell`/`_rlif_cell`/`_radlif_cell` methods."""

    @staticmethod
    def forward(ctx, kind, theta, Wx, alpha, beta, a, b, V, u0, w0, s0, steps_per_launch=None):
        _require_device(Wx, "Wx")
        Wx = _f32c(Wx)
        Bp, T, H = Wx.shape
        p = {k_: v for k_, v in dict(alpha=alpha, beta=beta, a=a, b=b, V=V).items() if v is not None}
        s, _, saved, _ = cell_forward(kind, Wx, None, None, p, u0, w0, s0, B=Bp, dirs=1, theta=theta, p_drop=0.0,
                                      seed=0, steps_per_launch=steps_per_launch, want_saves=any(ctx.needs_input_grad))
        ctx.kind, ctx.theta, ctx.dims, ctx.cell_saved, ctx.spl = kind, theta, (Bp, T, H), saved, steps_per_launch
        ctx.save_for_backward(alpha, beta, a, b, V, u0, w0, s0)
        return s

    @staticmethod
    def backward(ctx, g_s):
        alpha, beta, a, b, V, u0, w0, s0 = ctx.saved_tensors
        Bp, T, H = ctx.dims
        p = {k_: v for k_, v in dict(alpha=alpha, beta=beta, a=a, b=b, V=V).items() if v is not None}
        dWx, pg = cell_backward(ctx.kind, _f32c(g_s), None, p, u0, w0, s0, ctx.cell_saved, B=Bp, dirs=1, T=T,
                                H=H, theta=ctx.theta, p_drop=0.0, seed=0, steps_per_launch=ctx.spl)
        return (None, None, dWx, pg.get("alpha"), pg.get("beta"), pg.get("a"), pg.get("b"), pg.get("V"),
                None, None, None, None)


class ReadoutCellFn(torch.autograd.Function):
    """The readout cell in isolation (reference `_readout_cell`, snns.py:808-825)."""

    @staticmethod
    def forward(ctx, Wx, alpha, u0):
        _require_device(Wx, "Wx")
        Wx = _f32c(Wx)
        B, T, C = Wx.shape
        if C > 256:
            raise ValueError(f"sparch_amd: the readout kernels handle at most 256 classes (got {C})")
        out = torch.empty(B, C, dtype=torch.float32, device=Wx.device)
        u_save = torch.empty(B, T, C, dtype=torch.float32, device=Wx.device)
        check(lib.sparch_readout_fwd(B, T, C, ptr(Wx), None, None, ptr(alpha), ptr(u0), ptr(out), ptr(u_save),
                                     _stream()), "sparch_readout_fwd")
        ctx.dims = (B, T, C)
        ctx.save_for_backward(alpha, u0, u_save)
        return out

    @staticmethod
    def backward(ctx, g_out):
        alpha, u0, u_save = ctx.saved_tensors
        B, T, C = ctx.dims
        dWx = torch.empty(B, T, C, dtype=torch.float32, device=g_out.device)
        ws = torch.empty(1, B, C, dtype=torch.float32, device=g_out.device)
        check(lib.sparch_readout_bwd(B, T, C, ptr(_f32c(g_out)), None, None, None, ptr(u_save), ptr(alpha),
                                     ptr(u0), ptr(dWx), ptr(ws), _stream()), "sparch_readout_bwd")
        (dalpha,) = _finish_param_grads(ws, B, C, [alpha], [ALPHA_LIM])
        return dWx, dalpha, None


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss()(output, y) of the train step (exp.py:100, 362; mean reduction) with its gradient, in
    one launch (`sparch_ce_loss`): eager torch runs seven small kernels for it per step."""

    @staticmethod
    def forward(ctx, logits, labels):
        _require_device(logits, "logits")
        logits = _f32c(logits)
        B, C = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits)
        check(lib.sparch_ce_loss(B, C, ptr(logits), ptr(labels.contiguous()), ptr(loss), ptr(dlogits), _stream()),
              "sparch_ce_loss")
        ctx.save_for_backward(dlogits)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


def cross_entropy(logits, labels):
    """Mean cross-entropy of (B,C) logits against int64 class labels on the device (see CrossEntropyFn)."""
    if labels.dtype != torch.int64 or logits.ndim != 2:
        return torch.nn.functional.cross_entropy(logits, labels)
    return CrossEntropyFn.apply(logits, labels)


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for the reference's `nn.CrossEntropyLoss()` (exp.py:100) on the HIP path."""

    def forward(self, output, target):
        return cross_entropy(output, target)


def fbank(wave, num_mel_bins=40):
    """Kaldi-style log-mel filterbank of (clips, samples) fp32 audio on the device
    (replaces torchaudio.compliance.kaldi.fbank at nonspiking_datasets.py:96,194)."""
    _require_device(wave, "waveform")
    wave = _f32c(wave)
    if wave.ndim == 1:
        wave = wave[None]
    n_clips, n_samples = wave.shape
    frames = lib.sparch_fbank_frames(n_samples)
    out = torch.empty(n_clips, frames, num_mel_bins, dtype=torch.float32, device=wave.device)
    tok = timer.start(f"fbank[{n_clips}x{n_samples}]")
    check(lib.sparch_fbank_fwd(n_clips, n_samples, num_mel_bins, ptr(wave), ptr(out), _stream()),
          "sparch_fbank_fwd")
    timer.stop(tok)
    return out


def bin_events(times, units, nb_steps=100, nb_units=700, max_time=1.4, device="cuda"):
    """Batch of event lists -> dense (B, nb_steps, nb_units) float32 spike counts on the device, as the
    reference's SpikingDataset.__getitem__ builds them per sample on the CPU (spiking_datasets.py:66-78).
    times / units: sequences (one entry per sample) of 1-D arrays or tensors.  Returns (x, n_dropped) where
    n_dropped is a device int32 tensor counting events the reference would have rejected."""
    import numpy as np

    lens = [len(t) for t in times]
    if len(lens) == 0 or len(units) != len(lens):
        raise ValueError("bin_events: times and units must be non-empty sequences of equal length")
    offs = torch.zeros(len(lens) + 1, dtype=torch.int64)
    offs[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int64), 0)
    n = int(offs[-1])
    t_all = torch.from_numpy(np.concatenate([np.asarray(t, np.float32).ravel() for t in times]).astype(np.float32)) \
        if n else torch.zeros(0, dtype=torch.float32)
    u_all = torch.from_numpy(np.concatenate([np.asarray(u).ravel() for u in units]).astype(np.int32)) \
        if n else torch.zeros(0, dtype=torch.int32)
    dev = torch.device(device)
    t_d, u_d, o_d = t_all.to(dev), u_all.to(dev), offs.to(dev)
    out = torch.empty(len(lens), nb_steps, nb_units, dtype=torch.float32, device=dev)
    dropped = torch.empty(4, dtype=torch.int32, device=dev)
    check(lib.sparch_bin_events(n, ptr(t_d) if n else None, ptr(u_d) if n else None, ptr(o_d), len(lens), nb_steps,
                                nb_units, float(max_time), ptr(out), ptr(dropped), _stream()), "sparch_bin_events")
    return out, dropped[:1]


# ----------------------------------------------------------------------------- f-4: non-spiking baselines
ACT_KIND = {"sigmoid": 0, "relu": 1, "tanh": 2}


class MLPLayerFn(torch.autograd.Function):
    """x (B,T,K) -> dropout(act(norm(W x))) (B,T,H): MLPLayer.forward, anns.py:210-227."""

    @staticmethod
    def forward(ctx, cfg, x, W, Wb, nw, nb):
        _require_device(x, "input")
        _require_device(W, "layer parameters")
        norm, training = cfg["normalization"], cfg["training"]
        x = _f32c(x)
        B, T, K = x.shape
        H = W.shape[0]
        if H % 4 != 0:
            raise ValueError("sparch_amd: MLP layers need hidden_size % 4 == 0")
        M = B * T
        x2 = x.view(M, K)
        Wx_raw, colstat = gemm_nt(x2, W, Wb, colstat=(norm == "batchnorm" and training))          # anns.py:218
        Wx_in, scale, shift, nsaved = _Norm.forward(norm, Wx_raw, colstat, nw, nb, cfg.get("running_mean"),
                                                    cfg.get("running_var"), training, 1,          # 221-223
                                                    ln_width=cfg.get("ln_width"))
        y = torch.empty(B, T, H, dtype=torch.float32, device=x.device)
        check(lib.sparch_act_fwd(ACT_KIND[cfg["act"]], M * H, H, ptr(Wx_in), ptr(scale), ptr(shift),
                                 cfg["p_drop"], cfg["seed"], ptr(y), _stream()), "sparch_act_fwd")  # 226
        ctx.cfg, ctx.shape, ctx.nsaved = cfg, (B, T, K, H), nsaved
        ctx.save_for_backward(x2, W, nw, Wx_raw if norm in ("batchnorm", "layernorm") else None,
                              Wx_in if norm != "batchnorm" else None, scale, shift)
        return y

    @staticmethod
    def backward(ctx, g_y):
        cfg = ctx.cfg
        norm = cfg["normalization"]
        B, T, K, H = ctx.shape
        x2, W, nw, Wx_raw, Wx_in, scale, shift = ctx.saved_tensors
        M = B * T
        z = Wx_in if Wx_in is not None else Wx_raw
        dz = torch.empty(M, H, dtype=torch.float32, device=x2.device)
        check(lib.sparch_act_bwd(ACT_KIND[cfg["act"]], M * H, H, ptr(z), ptr(scale), ptr(shift), ptr(_f32c(g_y)),
                                 cfg["p_drop"], cfg["seed"], ptr(dz), _stream()), "sparch_act_bwd")
        dx_raw, dnw, dnb = _Norm.backward(norm, dz, Wx_raw, nw, ctx.nsaved, cfg["training"],
                                          ln_width=cfg.get("ln_width"))
        dW = gemm_tn(dx_raw, x2)
        dWb = _colsum(dx_raw) if ctx.needs_input_grad[3] else None
        dx = gemm_nn(dx_raw, W).view(B, T, K) if ctx.needs_input_grad[1] else None
        return None, dx, dW, dWb, dnw, dnb


class ReadoutANNFn(torch.autograd.Function):
    """x (B,T,K) -> norm(W sum_t softmax(x_t)) (B,C): ReadoutLayerANN.forward, anns.py:644-665."""

    @staticmethod
    def forward(ctx, cfg, x, W, Wb, nw, nb):
        _require_device(x, "input")
        _require_device(W, "layer parameters")
        norm, training = cfg["normalization"], cfg["training"]
        x = _f32c(x)
        B, T, K = x.shape
        C = W.shape[0]
        if K % 4 != 0 or K > 4096:
            raise ValueError("sparch_amd: ANN readout needs input features % 4 == 0 and <= 4096")
        y = torch.empty(B, K, dtype=torch.float32, device=x.device)
        check(lib.sparch_softmax_sum_fwd(B, T, K, ptr(x), ptr(y), _stream()), "sparch_softmax_sum_fwd")  # 658-663
        Wy_raw, colstat = gemm_nt(y, W, Wb, colstat=(norm == "batchnorm" and training))                  # 650
        Wy_in, scale, shift, nsaved = _Norm.forward(norm, Wy_raw, colstat, nw, nb, cfg.get("running_mean"),
                                                    cfg.get("running_var"), training, 1)                 # 653-654
        out = Wy_in if scale is None else Wy_in * scale + shift  # (B,C): tiny
        ctx.cfg, ctx.shape, ctx.nsaved = cfg, (B, T, K, C), nsaved
        ctx.save_for_backward(x, y, W, nw, Wy_raw if norm in ("batchnorm", "layernorm") else None, scale)
        return out

    @staticmethod
    def backward(ctx, g_out):
        cfg = ctx.cfg
        norm = cfg["normalization"]
        B, T, K, C = ctx.shape
        x, y, W, nw, Wy_raw, scale = ctx.saved_tensors
        dz = _f32c(g_out).clone()
        dx_raw, dnw, dnb = _Norm.backward(norm, dz, Wy_raw, nw, ctx.nsaved, cfg["training"])
        dW = gemm_tn(dx_raw, y)
        dWb = _colsum(dx_raw) if ctx.needs_input_grad[3] else None
        dx = None
        if ctx.needs_input_grad[1]:
            gy = gemm_nn(dx_raw, W)  # (B,K)
            dx = torch.empty(B, T, K, dtype=torch.float32, device=x.device)
            check(lib.sparch_softmax_sum_bwd(B, T, K, ptr(x), ptr(gy), ptr(dx), _stream()), "sparch_softmax_sum_bwd")
        return None, dx, dW, dWb, dnw, dnb


class RNNLayerFn(torch.autograd.Function):
    """x (B,T,K) -> dropout(y) (B,T,H*dirs) with y_t = act(norm(W x)_t + y_{t-1} V^T): RNNLayer.forward,
    anns.py:295-339.  The flipped copy of a bidirectional layer is never materialised (as for the spiking
    layers: W(x.flip(1)) = W(x).flip(1), BatchNorm over B'T rows = over BT rows)."""

    @staticmethod
    def forward(ctx, cfg, x, W, Wb, nw, nb, V):
        _require_device(x, "input")
        _require_device(W, "layer parameters")
        norm, training, dirs = cfg["normalization"], cfg["training"], cfg["dirs"]
        x = _f32c(x)
        B, T, K = x.shape
        H = W.shape[0]
        if H % 4 != 0:
            raise ValueError("sparch_amd: recurrent layers need hidden_size % 4 == 0")
        M = B * T
        dev = x.device
        x2 = x.view(M, K)
        Wx_raw, colstat = gemm_nt(x2, W, Wb, colstat=(norm == "batchnorm" and training))             # anns.py:306
        Wx_in, scale, shift, nsaved = _Norm.forward(norm, Wx_raw, colstat, nw, nb, cfg.get("running_mean"),
                                                    cfg.get("running_var"), training, dirs,          # 309-311
                                                    ln_width=cfg.get("ln_width"))
        Bp = B * dirs
        y_out = torch.empty(B, T, H * dirs, dtype=torch.float32, device=dev)
        y_state = torch.empty(Bp, T, H, dtype=torch.float32, device=dev)
        if rec_step_path(H):  # one launch per step, y_{t-1} V^T between the steps (exact six-term split GEMM)
            y_step = torch.empty(Bp, H, dtype=torch.float32, device=dev)
            rec = None
            tok = timer.start("ann_rec_fwd_steps[RNN]")
            for t in range(T):
                if t > 0:
                    rec, _ = gemm_nt(y_step, V)
                check(lib.sparch_ann_rec_step_fwd(ACT_KIND[cfg["act"]], B, dirs, T, H, t, ptr(Wx_in), ptr(scale),
                                                  ptr(shift), ptr(rec), cfg["p_drop"], cfg["seed"], ptr(y_out),
                                                  ptr(y_state), ptr(y_step), _stream()), "sparch_ann_rec_step_fwd")
            timer.stop(tok)
        else:
            vpack = torch.empty(lib.sparch_vpack_bytes(H) // 4, dtype=torch.float32, device=dev)
            check(lib.sparch_vpack(H, ptr(V), 1 | 2, ptr(vpack), None, _stream(), _prec()), "sparch_vpack")   # y V^T, dense
            nbytes = lib.sparch_rec_chan_bytes(Bp, T, H)
            chan = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)  # (ceil: the byte count need not be a multiple of 8)
            tok = timer.start("ann_rec_fwd[RNN]")
            with _persistent_launch():
                check(lib.sparch_ann_rec_fwd(ACT_KIND[cfg["act"]], B, dirs, T, H, ptr(Wx_in), ptr(scale), ptr(shift),
                                             ptr(vpack), cfg["p_drop"], cfg["seed"], ptr(y_out), ptr(y_state),
                                             ptr(chan), nbytes, ptr(status_word(dev)), rec_steps_per_launch(T),
                                             _stream()), "sparch_ann_rec_fwd")
            timer.stop(tok)
        ctx.cfg, ctx.shape, ctx.nsaved = cfg, (B, T, K, H), nsaved
        ctx.save_for_backward(x2, W, nw, V, y_state, Wx_raw if norm in ("batchnorm", "layernorm") else None)
        return y_out

    @staticmethod
    def backward(ctx, g_y):
        cfg = ctx.cfg
        norm, dirs = cfg["normalization"], cfg["dirs"]
        B, T, K, H = ctx.shape
        x2, W, nw, V, y_state, Wx_raw = ctx.saved_tensors
        M, Bp = B * T, B * dirs
        dev = x2.device
        dpre = torch.empty(Bp, T, H, dtype=torch.float32, device=dev)
        y_prev = torch.empty(Bp, T, H, dtype=torch.float32, device=dev)
        g_y = _f32c(g_y)
        if rec_step_path(H):
            dpre_step = torch.empty(Bp, H, dtype=torch.float32, device=dev)
            rec = None
            tok = timer.start("ann_rec_bwd_steps[RNN]")
            for s_ in range(T):
                if s_ > 0:
                    rec = gemm_nn(dpre_step, V)           # dpre_{t+1} V
                check(lib.sparch_ann_rec_step_bwd(ACT_KIND[cfg["act"]], B, dirs, T, H, s_, ptr(g_y), ptr(y_state),
                                                  ptr(rec), cfg["p_drop"], cfg["seed"], ptr(dpre), ptr(y_prev),
                                                  ptr(dpre_step), _stream()), "sparch_ann_rec_step_bwd")
            timer.stop(tok)
        else:
            vpack = torch.empty(lib.sparch_vpack_bytes(H) // 4, dtype=torch.float32, device=dev)
            check(lib.sparch_vpack(H, ptr(V), 0 | 2, ptr(vpack), None, _stream(), _prec()), "sparch_vpack")   # dpre V, dense
            nbytes = lib.sparch_rec_chan_bytes(Bp, T, H)
            chan = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)  # (ceil: the byte count need not be a multiple of 8)
            tok = timer.start("ann_rec_bwd[RNN]")
            with _persistent_launch():
                check(lib.sparch_ann_rec_bwd(ACT_KIND[cfg["act"]], B, dirs, T, H, ptr(g_y), ptr(y_state), ptr(vpack),
                                             cfg["p_drop"], cfg["seed"], ptr(dpre), ptr(y_prev), ptr(chan), nbytes,
                                             ptr(status_word(dev)), rec_steps_per_launch(T), _stream()),
                      "sparch_ann_rec_bwd")
            timer.stop(tok)
        # dV[i][j] = sum over rows and steps of dpre[.,i] * y_{t-1}[.,j]   (V(y) = y V^T, anns.py:336)
        dV = gemm_tn(dpre.view(Bp * T, H), y_prev.view(Bp * T, H))
        if dirs == 2:  # both directions share the projection rows (anns.py:298-300)
            dy = torch.empty(B, T, H, dtype=torch.float32, device=dev)
            check(lib.sparch_add_halves(M * H, ptr(dpre), ptr(dy), _stream()), "sparch_add_halves")
        else:
            dy = dpre
        dx_raw, dnw, dnb = _Norm.backward(norm, dy.view(M, H), Wx_raw, nw, ctx.nsaved, cfg["training"],
                                          ln_width=cfg.get("ln_width"))
        dW = gemm_tn(dx_raw, x2)
        dWb = _colsum(dx_raw) if ctx.needs_input_grad[3] else None
        dx = gemm_nn(dx_raw, W).view(B, T, K) if ctx.needs_input_grad[1] else None
        return None, dx, dW, dWb, dnw, dnb, dV


def _gemm_small(A, B, nn):
    """Per-step recurrent product of the gated baselines: A (M,K) @ B^T (nn=False, B (N,K)) or @ B (nn=True,
    B (K,N)) with M*N small and K long -> split-K form (fills the GPU instead of 16 workgroups)."""
    M, K = A.shape
    N = B.shape[1] if nn else B.shape[0]
    C = torch.empty(M, N, dtype=torch.float32, device=A.device)
    nbytes = lib.sparch_gemm6_splitk_workspace_bytes(M, N, K, _prec())
    ws = torch.empty(max(nbytes, 16) // 4, dtype=torch.float32, device=A.device)
    fn = lib.sparch_gemm6_nn_splitk if nn else lib.sparch_gemm6_nt_splitk
    check(fn(M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), N, ptr(ws), nbytes, _stream(), _prec()),
          "sparch_gemm6_splitk")
    return C


def _gate_step(mode, B, dirs, T, H, t, ins, outs, p_drop, seed):
    """One launch of sparch_gate_step; ins / outs: dicts slot name -> tensor (missing = NULL)."""
    in_names = ["Wx", "sc", "sh", "Wzx", "scz", "shz", "Wrx", "scr", "shr", "rec", "g_out", "carry_mv", "carry_dir", "dry"]
    out_names = ["y_state", "z_save", "r_save", "c_save", "ry", "y_out", "carry_dir_out", "dgate", "dcp", "dz_all",
                 "dr_all", "dc_all", "yprev_all", "ry_all"]
    arr = ctypes.c_void_p * 14
    a_in = arr(*[(ins[n].data_ptr() if ins.get(n) is not None else None) for n in in_names])
    a_out = arr(*[(outs[n].data_ptr() if outs.get(n) is not None else None) for n in out_names])
    check(lib.sparch_gate_step(mode, B, dirs, T, H, t, a_in, a_out, p_drop, seed, _stream()), "sparch_gate_step")


class GatedLayerFn(torch.autograd.Function):
    """LiGRU / GRU baseline layers (anns.py:412-462, 540-595): x (B,T,K) -> dropout(y) (B,T,H*dirs).
    Launch-per-step this round: per time step the recurrent products run on the exact-split GEMMs and the
    gate arithmetic in `sparch_gate_step`; projections, normalisation and all weight gradients are whole-
    sequence GEMMs as for the other layers.  mats = ("c", "z") for LiGRU, ("c", "z", "r") for GRU; the tensor
    arguments come in groups (W, Wb, norm weight, norm bias, V) per matrix in that order."""

    @staticmethod
    def forward(ctx, cfg, x, *params):
        _require_device(x, "input")
        kind, norm, training, dirs = cfg["kind"], cfg["normalization"], cfg["training"], cfg["dirs"]
        mats = ("c", "z", "r") if kind == "GRU" else ("c", "z")
        P = {m: dict(zip(("W", "Wb", "nw", "nb", "V"), params[5 * i:5 * i + 5])) for i, m in enumerate(mats)}
        x = _f32c(x)
        B, T, K = x.shape
        H = P["c"]["W"].shape[0]
        if H % 4 != 0:
            raise ValueError("sparch_amd: recurrent layers need hidden_size % 4 == 0")
        M, Bp, dev = B * T, B * dirs, x.device
        x2 = x.view(M, K)
        proj = {}
        for m in mats:
            raw, colstat = gemm_nt(x2, P[m]["W"], P[m]["Wb"], colstat=(norm == "batchnorm" and training))
            z_in, sc, sh, nsaved = _Norm.forward(norm, raw, colstat, P[m]["nw"], P[m]["nb"], cfg["running"][m][0],
                                                 cfg["running"][m][1], training, dirs, ln_width=cfg.get("ln_width"))
            proj[m] = dict(raw=raw, z_in=z_in, sc=sc, sh=sh, nsaved=nsaved)
        Vgate = torch.cat([P["z"]["V"], P["r"]["V"] if kind == "GRU" else P["c"]["V"]], dim=0).contiguous()  # (2H,H)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        y_state, z_save, c_save = new(Bp, T, H), new(Bp, T, H), new(Bp, T, H)
        r_save = new(Bp, T, H) if kind == "GRU" else None
        ry = new(Bp, H) if kind == "GRU" else None
        y_out = new(B, T, H * dirs)
        ins = {"Wx": proj["c"]["z_in"], "sc": proj["c"]["sc"], "sh": proj["c"]["sh"],
               "Wzx": proj["z"]["z_in"], "scz": proj["z"]["sc"], "shz": proj["z"]["sh"]}
        if kind == "GRU":
            ins.update(Wrx=proj["r"]["z_in"], scr=proj["r"]["sc"], shr=proj["r"]["sh"])
        outs = {"y_state": y_state, "z_save": z_save, "r_save": r_save, "c_save": c_save, "ry": ry, "y_out": y_out}
        p_drop, seed = cfg["p_drop"], cfg["seed"]
        persistent = kind == "LiGRU" and ligru_persistent_ok(H)
        ctx.gru_persistent = kind == "GRU" and gru_persistent_ok(H)
        if ctx.gru_persistent:
            # both hand-offs of a step inside one persistent launch per row-tile group (gatedcell.hip)
            vg = torch.empty(lib.sparch_gru_vpack_bytes(H, 0, 0) // 4, dtype=torch.float32, device=dev)
            vc = torch.empty(lib.sparch_gru_vpack_bytes(H, 0, 1) // 4, dtype=torch.float32, device=dev)
            check(lib.sparch_gru_vpack(H, ptr(P["z"]["V"]), ptr(P["r"]["V"]), ptr(P["c"]["V"]), 0, ptr(vg), ptr(vc),
                                       _stream(), _prec()), "sparch_gru_vpack")
            nbytes = lib.sparch_gru_chan_bytes(Bp, H)
            chan = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)  # (ceil: the byte count need not be a multiple of 8)
            tok = timer.start("gru_fwd")
            with _persistent_launch():
                check(lib.sparch_gru_fwd(B, dirs, T, H, ptr(proj["c"]["z_in"]), ptr(proj["c"]["sc"]), ptr(proj["c"]["sh"]),
                                         ptr(proj["z"]["z_in"]), ptr(proj["z"]["sc"]), ptr(proj["z"]["sh"]),
                                         ptr(proj["r"]["z_in"]), ptr(proj["r"]["sc"]), ptr(proj["r"]["sh"]), ptr(vg), ptr(vc),
                                         p_drop, seed, ptr(y_out), ptr(y_state), ptr(z_save), ptr(r_save), ptr(c_save),
                                         ptr(chan), nbytes, ptr(status_word(dev)), rec_steps_per_launch(T), _stream()),
                      "sparch_gru_fwd")
            timer.stop(tok)
        elif persistent:
            # the whole time loop in one persistent launch per row-tile group (gatedcell.hip)
            vp = torch.empty(lib.sparch_ligru_vpack_bytes(H, 0) // 4, dtype=torch.float32, device=dev)
            check(lib.sparch_ligru_vpack(H, ptr(P["z"]["V"]), ptr(P["c"]["V"]), 0, ptr(vp), _stream(), _prec()), "sparch_ligru_vpack")
            nbytes = lib.sparch_ligru_chan_bytes(Bp, H)
            chan = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)  # (ceil: the byte count need not be a multiple of 8)
            tok = timer.start("ligru_fwd")
            with _persistent_launch():
                check(lib.sparch_ligru_fwd(B, dirs, T, H, ptr(proj["c"]["z_in"]), ptr(proj["c"]["sc"]), ptr(proj["c"]["sh"]),
                                           ptr(proj["z"]["z_in"]), ptr(proj["z"]["sc"]), ptr(proj["z"]["sh"]), ptr(vp),
                                           p_drop, seed, ptr(y_out), ptr(y_state), ptr(z_save), ptr(c_save), ptr(chan),
                                           nbytes, ptr(status_word(dev)), rec_steps_per_launch(T), _stream()),
                      "sparch_ligru_fwd")
            timer.stop(tok)
        else:
            tok = timer.start(f"gated_fwd[{kind}]")
            for t in range(T):
                rec = _gemm_small(y_state[:, t - 1, :], Vgate, nn=False) if t > 0 else None   # y_{t-1} [Vz;V]^T  (anns.py:457-458)
                if kind == "LiGRU":
                    _gate_step(0, B, dirs, T, H, t, dict(ins, rec=rec), outs, p_drop, seed)
                else:
                    _gate_step(1, B, dirs, T, H, t, dict(ins, rec=rec), outs, p_drop, seed)
                    recc = _gemm_small(ry, P["c"]["V"], nn=False) if t > 0 else None       # (r y_{t-1}) V^T  (anns.py:591)
                    _gate_step(2, B, dirs, T, H, t, dict(ins, rec=recc), outs, p_drop, seed)
            timer.stop(tok)
        ctx.cfg, ctx.shape, ctx.mats = cfg, (B, T, K, H), mats
        ctx.nsaved = {m: proj[m]["nsaved"] for m in mats}
        ctx.needs_b = {m: P[m]["Wb"] is not None for m in mats}
        saved = [x2, y_state, z_save, c_save, r_save, Vgate]
        for m in mats:
            saved += [P[m]["W"], P[m]["nw"], P[m]["V"], proj[m]["raw"] if norm in ("batchnorm", "layernorm") else None]
        ctx.save_for_backward(*saved)
        return y_out

    @staticmethod
    def backward(ctx, g_y):
        cfg, mats = ctx.cfg, ctx.mats
        kind, norm, dirs = cfg["kind"], cfg["normalization"], cfg["dirs"]
        B, T, K, H = ctx.shape
        sv = ctx.saved_tensors
        x2, y_state, z_save, c_save, r_save, Vgate = sv[:6]
        Pm = {m: dict(zip(("W", "nw", "V", "raw"), sv[6 + 4 * i:10 + 4 * i])) for i, m in enumerate(mats)}
        M, Bp, dev = B * T, B * dirs, x2.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        d_all = {"z": new(Bp, T, H), "c": new(Bp, T, H)}
        yprev_all = new(Bp, T, H)
        ry_all = new(Bp, T, H) if kind == "GRU" else None
        if kind == "GRU":
            d_all["r"] = new(Bp, T, H)
        dgate, dcp = new(Bp, 2 * H), (new(Bp, H) if kind == "GRU" else None)
        cdir = [new(Bp, H), new(Bp, H)]
        ins = {"g_out": _f32c(g_y)}
        outs = {"y_state": y_state, "z_save": z_save, "r_save": r_save, "c_save": c_save, "dgate": dgate, "dcp": dcp,
                "dz_all": d_all["z"], "dc_all": d_all["c"], "dr_all": d_all.get("r"), "yprev_all": yprev_all,
                "ry_all": ry_all}
        p_drop, seed = cfg["p_drop"], cfg["seed"]
        if kind == "GRU" and ctx.gru_persistent and gru_persistent_ok(H):
            vg = torch.empty(lib.sparch_gru_vpack_bytes(H, 1, 0) // 4, dtype=torch.float32, device=dev)
            vc = torch.empty(lib.sparch_gru_vpack_bytes(H, 1, 1) // 4, dtype=torch.float32, device=dev)
            check(lib.sparch_gru_vpack(H, ptr(Pm["z"]["V"]), ptr(Pm["r"]["V"]), ptr(Pm["c"]["V"]), 1, ptr(vg), ptr(vc),
                                       _stream(), _prec()), "sparch_gru_vpack")
            nbytes = lib.sparch_gru_chan_bytes(Bp, H)
            chan = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)  # (ceil: the byte count need not be a multiple of 8)
            carry = new(Bp, H)
            tok = timer.start("gru_bwd")
            with _persistent_launch():
                check(lib.sparch_gru_bwd(B, dirs, T, H, ptr(_f32c(g_y)), ptr(y_state), ptr(z_save), ptr(r_save), ptr(c_save),
                                         ptr(vg), ptr(vc), p_drop, seed, ptr(d_all["z"]), ptr(d_all["r"]), ptr(d_all["c"]),
                                         ptr(yprev_all), ptr(ry_all), ptr(carry), ptr(chan), nbytes, ptr(status_word(dev)),
                                         rec_steps_per_launch(T), _stream()), "sparch_gru_bwd")
            timer.stop(tok)
        elif kind == "LiGRU" and ligru_persistent_ok(H):
            vpb = torch.empty(lib.sparch_ligru_vpack_bytes(H, 1) // 4, dtype=torch.float32, device=dev)
            check(lib.sparch_ligru_vpack(H, ptr(Pm["z"]["V"]), ptr(Pm["c"]["V"]), 1, ptr(vpb), _stream(), _prec()), "sparch_ligru_vpack")
            nbytes = lib.sparch_ligru_chan_bytes(Bp, H)
            chan = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)  # (ceil: the byte count need not be a multiple of 8)
            carry = new(Bp, H)
            tok = timer.start("ligru_bwd")
            with _persistent_launch():
                check(lib.sparch_ligru_bwd(B, dirs, T, H, ptr(_f32c(g_y)), ptr(y_state), ptr(z_save), ptr(c_save), ptr(vpb),
                                           p_drop, seed, ptr(d_all["z"]), ptr(d_all["c"]), ptr(yprev_all), ptr(carry),
                                           ptr(chan), nbytes, ptr(status_word(dev)), rec_steps_per_launch(T), _stream()),
                      "sparch_ligru_bwd")
            timer.stop(tok)
        else:
            carry_mv = carry_dir = None
            tok = timer.start(f"gated_bwd[{kind}]")
            for t in range(T - 1, -1, -1):
                o = dict(outs, carry_dir_out=cdir[t & 1])
                i = dict(ins, carry_mv=carry_mv, carry_dir=carry_dir)
                if kind == "LiGRU":
                    _gate_step(3, B, dirs, T, H, t, i, o, p_drop, seed)
                else:
                    _gate_step(4, B, dirs, T, H, t, i, o, p_drop, seed)
                    dry = _gemm_small(dcp, Pm["c"]["V"], nn=True)        # gradient of r * y_{t-1}
                    _gate_step(5, B, dirs, T, H, t, {"dry": dry}, o, p_drop, seed)
                if t > 0:
                    carry_mv = _gemm_small(dgate, Vgate, nn=True)        # [dz_pre | d*_pre] [Vz; V*]
                    carry_dir = cdir[t & 1]
            timer.stop(tok)
        flat = lambda a: a.view(Bp * T, H)  # noqa: E731
        dV = {"z": gemm_tn(flat(d_all["z"]), flat(yprev_all))}
        if kind == "GRU":
            dV["r"] = gemm_tn(flat(d_all["r"]), flat(yprev_all))
            dV["c"] = gemm_tn(flat(d_all["c"]), flat(ry_all))
        else:
            dV["c"] = gemm_tn(flat(d_all["c"]), flat(yprev_all))
        grads, dx_raws = {}, []
        for m in mats:
            if dirs == 2:  # both directions share the projection rows
                dy = new(B, T, H)
                check(lib.sparch_add_halves(M * H, ptr(d_all[m]), ptr(dy), _stream()), "sparch_add_halves")
            else:
                dy = d_all[m]
            dx_raw, dnw, dnb = _Norm.backward(norm, dy.view(M, H), Pm[m]["raw"], Pm[m]["nw"], ctx.nsaved[m],
                                              cfg["training"], ln_width=cfg.get("ln_width"))
            dx_raws.append(dx_raw)
            grads[m] = (gemm_tn(dx_raw, x2), _colsum(dx_raw) if ctx.needs_b[m] else None, dnw, dnb, dV[m])
        dx = None
        if ctx.needs_input_grad[1]:
            dx = gemm_nn(torch.cat(dx_raws, dim=1), torch.cat([Pm[m]["W"] for m in mats], dim=0)).view(B, T, K)
        out = [None, dx]
        for m in mats:
            out += list(grads[m])
        return tuple(out)
