"""
Drop-in mirror of the reference's `sparch.models.snns` module API, backed by the
MI355X HIP library.

Same public names, constructor signatures, attribute / parameter / state_dict names,
construction-time RNG draw order and per-forward initial-state draw order as
/root/reference/sparch/models/snns.py (SNN 39-176, LIFLayer 179-303, adLIFLayer 306-445,
RLIFLayer 448-578, RadLIFLayer 581-727, ReadoutLayer 730-825, SpikeFunctionBoxcar 20-36),
so checkpoints, `Experiment` and user code keep working.  What differs is below the API:
a layer's forward is one autograd node (`functional.SpikingLayerFn`) that runs the
projection GEMM, the normalisation and the whole T-step cell recurrence as HIP kernels,
and `SNN.forward` obtains firing rates from spike counts the cell kernels already made
instead of materialising `cat(all_spikes)` (snns.py:174).

`sparch.models.snns` (repo root) re-exports this module under the reference's import path.
"""
import ctypes
import math
import os

import torch
import torch.nn as nn

from . import functional as Fn

_SPIKING_KINDS = ("LIF", "adLIF", "RLIF", "RadLIF")


class SpikeFunctionBoxcar(torch.autograd.Function):
    """Heaviside step with a box-car surrogate gradient (reference snns.py:20-36):
    forward `x > 0` (strict), backward passes the gradient where -0.5 < x <= 0.5.
    Kept for API compatibility; inside the layers this pair is fused into the cell kernels."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return (x > 0).to(torch.float32)

    @staticmethod
    def backward(ctx, grad_spikes):
        # the reference ASSIGNS zeros outside the box (snns.py:33-35): a non-finite upstream gradient is cleared
        # there, and a NaN x (both comparisons false) lets the gradient through — as the cell kernels do
        # (csrc/common.h boxcar_gate) and oracle/snn_oracle.py
        (x,) = ctx.saved_tensors
        grad_x = grad_spikes.clone()
        grad_x[x <= -0.5] = 0
        grad_x[x > 0.5] = 0
        return grad_x


def _tag_spikes(s, scale, s16, placeholder=False):
    """Mark `s` as a spike train of ours (entries 0 or `scale`, `s16` the same spikes as a bf16 0/1 plane) so
    that the next layer can take the exact spike GEMMs.  The tag carries the tensor's version counter: an
    in-place edit between the layers (mul_, masked fill, slice assignment) invalidates it.
    placeholder: `s` holds no values (forward_with_rate(fp32_out=False)); the plane is the data."""
    s._sparch_spike_tag = (s._version, tuple(s.shape), float(scale), s16, bool(placeholder))


def materialize_spikes(s):
    """The values of a spiking layer's output as an fp32 tensor: `s` itself, or — for the placeholder
    SNN.forward passes between its own layers — its bf16 plane times the dropout scale (the same numbers the
    reference's tensor holds, snns.py:278)."""
    tag = getattr(s, "_sparch_spike_tag", None)
    if tag is not None and len(tag) > 4 and tag[4]:
        return tag[3].float() * tag[2]
    return s


def _spike_tag(x):
    """(scale, s16) if x still is the untouched output of one of our spiking layers, else (None, None):
    the layer then treats x as an arbitrary real-valued input (device-gated exact path / dense GEMM)."""
    tag = getattr(x, "_sparch_spike_tag", None)
    if tag is None or tag[0] != x._version or tag[1] != tuple(x.shape):
        return None, None
    return tag[2], tag[3]


# ---- initial-state draws.  The reference draws every layer's u0, [w0,] s0 with torch.rand from the global CPU
# generator at each forward (snns.py:286-287, 423-425, 558-559, 700-702, 812).  The values below are the same
# numbers from the same generator in the same order; what differs is who computes them: torch's serial loop
# costs ~5 ns per number (8.5 ms per step at the headline shape, more than the GPU needs for the step), the
# library's host routine `sparch_mt19937_uniform_f32` ~1 ns.  It works on the generator's serialized MT19937
# state (CPUGeneratorImplStateLegacy: seed u64, left i32, seeded i32, next u64, 624 state words as u64, ...),
# which is moved out, advanced, and written back — so anything else drawing from the global generator before or
# after sees the stream it would have seen.  The layout is verified once against torch.rand on a private
# generator; if that check ever fails (another torch build), the draws simply stay with torch.rand.
_MT_WORDS, _MT_OFF_LEFT, _MT_OFF_NEXT, _MT_OFF_KEY, _MT_STATE_BYTES = 624, 8, 16, 24, 24 + 624 * 8


def _mt_draw(state_bytes, outs):
    """Fill the fp32 CPU tensors `outs` (contiguous) in order from the serialized generator state
    `state_bytes` (uint8 tensor, modified in place).  Returns False if the state is not one this code reads."""
    import numpy as np

    raw = state_bytes.numpy()
    if raw.size < _MT_STATE_BYTES:
        return False
    left = int(raw[_MT_OFF_LEFT:_MT_OFF_LEFT + 4].view(np.int32)[0])
    nxt = int(raw[_MT_OFF_NEXT:_MT_OFF_NEXT + 8].view(np.uint64)[0])
    key64 = raw[_MT_OFF_KEY:_MT_OFF_KEY + _MT_WORDS * 8].view(np.uint64)
    if not (1 <= left <= _MT_WORDS and nxt <= _MT_WORDS and (left == 1 or left == _MT_WORDS + 1 - nxt)):
        return False
    key = key64.astype(np.uint32)
    pos = ctypes.c_int(_MT_WORDS if left == 1 else nxt)
    for t in outs:
        Fn.check(Fn.lib.sparch_mt19937_uniform_f32(key.ctypes.data, ctypes.byref(pos), t.numel(), t.data_ptr()),
                 "sparch_mt19937_uniform_f32")
    key64[:] = key
    p = pos.value
    raw[_MT_OFF_NEXT:_MT_OFF_NEXT + 8].view(np.uint64)[0] = p
    raw[_MT_OFF_LEFT:_MT_OFF_LEFT + 4].view(np.int32)[0] = 1 if p == _MT_WORDS else _MT_WORDS + 1 - p
    return True


_fast_rand_ok = None


def _fast_rand_available():
    """One-time check of the serialized-state layout: 300 k numbers across many block boundaries from a private
    generator, by torch.rand and by the host routine; the generator must also end in the same state."""
    global _fast_rand_ok
    if _fast_rand_ok is None:
        ok = False
        if os.environ.get("SPARCH_FAST_RAND", "1") != "0":
            try:
                g = torch.Generator()
                g.manual_seed(0x5eed)
                torch.rand(7, generator=g)  # somewhere inside a block
                st = g.get_state().clone()
                want = [torch.rand(300, 1001, generator=g), torch.rand(5, generator=g)]
                got = [torch.empty(300, 1001), torch.empty(5)]
                ok = _mt_draw(st, got) and all(torch.equal(a, b) for a, b in zip(want, got)) \
                    and torch.equal(st, g.get_state())
            except Exception:  # any surprise in the layout: keep torch.rand
                ok = False
        _fast_rand_ok = ok
    return _fast_rand_ok


def _rand_batch(shapes, device, out=None):
    """torch.rand(*shape) for every shape, in order, from the global CPU generator — as ONE pinned staging
    buffer and one asynchronous copy; returns the device tensors (views of one allocation, 256-byte aligned).
    out: the flat device buffer of an earlier call with the same shapes (`.flat` of its first tensor's list):
    refilled in place — the static inputs of a captured step."""
    if _rand_to is not _rand_to_default:  # a caller substituted the per-tensor draw (tests inject fixture states)
        return _StateList([_rand_to(r, c, device) for r, c in shapes], None)
    sizes = [int(r) * int(c) for r, c in shapes]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 63) // 64 * 64
    on_gpu = device.type == "cuda"
    slot = _staging_slot(total, device) if on_gpu else None
    host = slot[0] if on_gpu else torch.empty(total, dtype=torch.float32)
    views = [host[o:o + n].view(r, c) for o, n, (r, c) in zip(offs, sizes, shapes)]
    done = False
    if _fast_rand_available():
        st = torch.get_rng_state()
        done = _mt_draw(st, views)
        if done:
            torch.set_rng_state(st)
    if not done:
        for v in views:
            torch.rand(v.shape[0], v.shape[1], out=v)
    ready = None
    if out is not None and on_gpu:
        # static states of a captured step: the upload runs on the side stream into one of two device staging
        # buffers — under whatever replay the GPU is still busy with — and only a device-to-device copy (7 MB at the
        # headline shape: ~10 us against ~150 us over PCIe) sits in stream order in front of the next replay
        assert out.numel() == total and out.device == device
        main = torch.cuda.current_stream(device)
        side = _upload_stream(device)
        dslot = _device_staging_slot(total, device)
        if dslot[1] is not None:
            side.wait_event(dslot[1])  # the copy that last read this staging buffer
        with torch.cuda.stream(side):
            dslot[0].copy_(host, non_blocking=True)
            up = torch.cuda.Event()
            up.record(side)
        slot[1] = up
        main.wait_event(up)
        out.copy_(dslot[0], non_blocking=True)
        dslot[1] = torch.cuda.Event()
        dslot[1].record(main)
        dev = out
    elif out is not None:
        assert out.numel() == total and out.device == device
        out.copy_(host)
        dev = out
    elif on_gpu:
        # The upload (7 MB at the headline shape: ~150 us on the DMA engine) goes on a side stream, so that it
        # runs under whatever the compute stream does before it first needs a state (the first layer's
        # projection): in stream order it sat between two steps with the GPU idle.  `.ready` is the event
        # the consumer waits for (SNN.forward hands it to the first layer's cell call).
        main = torch.cuda.current_stream(device)
        side = _upload_stream(device)
        with torch.cuda.stream(side):
            dev = host.to(device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(side)
            slot[1] = ready
        dev.record_stream(main)  # allocated on the side stream's pool, consumed on the compute stream
    else:
        dev = host.to(device)
    return _StateList([dev[o:o + n].view(r, c) for o, n, (r, c) in zip(offs, sizes, shapes)], dev, ready)


_upload_streams = {}
# Pinned staging buffers of the draws: a ring of four per (size, device), each reused once the copy that last read it
# has completed (its event).  A fresh `torch.empty(pin_memory=True)` per step goes to hipHostMalloc whenever the host
# runs ahead of the GPU by more blocks than the caching allocator has seen freed — one ~1 ms call per step for the
# first dozens of replays of a captured step (BASELINE configs[1], 20 timed steps: 1.55 ms per step against 1.0 ms).
_staging = {}
_STAGING_DEPTH = 4


_device_staging = {}


def _device_staging_slot(total, device):
    """[buffer, event of the device-to-device copy that last read it] — two per (size, device), alternating."""
    key = (int(total), device.index if device.index is not None else torch.cuda.current_device())
    ring = _device_staging.get(key)
    if ring is None:
        if len(_device_staging) >= 4:
            _device_staging.clear()
        ring = _device_staging[key] = {"next": 0, "slots": [[torch.empty(total, dtype=torch.float32, device=device), None]
                                                             for _ in range(2)]}
    slot = ring["slots"][ring["next"]]
    ring["next"] ^= 1
    return slot


def _staging_slot(total, device):
    key = (int(total), device.type, device.index if device.index is not None else torch.cuda.current_device())
    ring = _staging.get(key)
    if ring is None:
        if len(_staging) >= 8:  # shapes come and go (tests, variable batch sizes): keep the pinned footprint bounded
            _staging.clear()
        ring = _staging[key] = {"next": 0, "slots": [[torch.empty(total, dtype=torch.float32, pin_memory=True), None]
                                                      for _ in range(_STAGING_DEPTH)]}
    slot = ring["slots"][ring["next"]]
    ring["next"] = (ring["next"] + 1) % _STAGING_DEPTH
    if slot[1] is not None:
        slot[1].synchronize()  # the host is four draws ahead of the GPU: wait for the oldest upload
        slot[1] = None
    return slot


def _upload_stream(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _upload_streams:
        _upload_streams[key] = torch.cuda.Stream(device=device)
    return _upload_streams[key]


class _StateList(list):
    """The tensors of one _rand_batch call; `.flat` is the single device allocation they are views of, `.ready`
    the event of their upload (None: already ordered on the compute stream)."""

    def __init__(self, tensors, flat, ready=None):
        super().__init__(tensors)
        self.flat = flat
        self.ready = ready

    def wait_ready(self):
        """Make the current stream wait for the upload (no-op when it was stream-ordered)."""
        if self.ready is not None:
            torch.cuda.current_stream().wait_event(self.ready)
            self.ready = None


def _rand_to(rows, cols, device):
    """torch.rand from the global CPU generator (the reference's source of initial states), staged
    in pinned memory and copied asynchronously so the host does not stall once per layer."""
    batch = _rand_batch([(rows, cols)], device)
    batch.wait_ready()
    return batch[0]


_rand_to_default = _rand_to


def _make_norm(normalization, hidden_size):
    """BatchNorm1d(momentum=0.05) / LayerNorm / nothing (reference snns.py:238-244).  The
    torch modules only hold the parameters and running statistics; the maths runs in HIP."""
    if normalization == "batchnorm":
        return nn.BatchNorm1d(hidden_size, momentum=Fn.BN_MOMENTUM), True
    if normalization == "layernorm":
        return nn.LayerNorm(hidden_size), True
    return None, False


class _SpikingLayer(nn.Module):
    """Shared implementation of the four spiking layer types; `kind` selects adaptation
    (adLIF, RadLIF) and layer-wise recurrence (RLIF, RadLIF)."""

    kind = None  # set by subclasses

    def __init__(self, input_size, hidden_size, batch_size, threshold=1.0, dropout=0.0,
                 normalization="batchnorm", use_bias=False, bidirectional=False):
        super().__init__()
        adaptive = self.kind in ("adLIF", "RadLIF")
        recurrent = self.kind in ("RLIF", "RadLIF")

        self.input_size = int(input_size)
        self.hidden_size = int(hidden_size)
        self.threshold = threshold
        self.dropout = dropout
        self.normalization = normalization
        self.use_bias = use_bias
        self.bidirectional = bidirectional
        self.batch_size = batch_size * (1 + self.bidirectional)
        self.alpha_lim = [math.exp(-1 / 5), math.exp(-1 / 25)]
        if adaptive:
            self.beta_lim = [math.exp(-1 / 30), math.exp(-1 / 120)]
            self.a_lim = [-1.0, 1.0]
            self.b_lim = [0.0, 2.0]
        self.spike_fct = SpikeFunctionBoxcar.apply

        # Parameters, created and initialised in the reference's order so that identical
        # seeds give identical initial weights (snns.py:233-235, 363-372, 502-507, 638-649).
        self.W = nn.Linear(self.input_size, self.hidden_size, bias=use_bias)
        if recurrent:
            self.V = nn.Linear(self.hidden_size, self.hidden_size, bias=False)
        self.alpha = nn.Parameter(torch.empty(self.hidden_size))
        if adaptive:
            self.beta = nn.Parameter(torch.empty(self.hidden_size))
            self.a = nn.Parameter(torch.empty(self.hidden_size))
            self.b = nn.Parameter(torch.empty(self.hidden_size))
        nn.init.uniform_(self.alpha, self.alpha_lim[0], self.alpha_lim[1])
        if adaptive:
            nn.init.uniform_(self.beta, self.beta_lim[0], self.beta_lim[1])
            nn.init.uniform_(self.a, self.a_lim[0], self.a_lim[1])
            nn.init.uniform_(self.b, self.b_lim[0], self.b_lim[1])
        if recurrent:
            nn.init.orthogonal_(self.V.weight)

        norm, self.normalize = _make_norm(normalization, self.hidden_size)
        if norm is not None:
            self.norm = norm
        self.drop = nn.Dropout(p=dropout)
        self._calls = 0
        self._layer_index = 0  # set by SNN; decorrelates dropout masks between layers

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_seed_word", None)  # graph mode's device-side dropout seed: a step's state, not the network's
        state.pop("_seed_word_owner_advances", None)
        return state

    # ------------------------------------------------------------------ helpers
    @property
    def uses_persistent_kernel(self):
        """True if this layer's time loop runs as ONE persistent launch whose workgroups wait for each other
        (RLIF / RadLIF up to 1024 hidden units) — what sparch_amd.dp keys its all-reduce policy on."""
        return self.kind in ("RLIF", "RadLIF") and not Fn.rec_step_path(self.hidden_size)

    def _draw_states(self, rows, device):
        """Random initial u, [w], s from torch's global CPU generator, in the reference's
        order (snns.py:286-287, 423-425, 558-559, 700-702), then moved to the device."""
        H = self.hidden_size
        u0 = _rand_to(rows, H, device)
        w0 = _rand_to(rows, H, device) if self.kind in ("adLIF", "RadLIF") else None
        s0 = _rand_to(rows, H, device)
        return u0, w0, s0

    def _dropout_seed(self, device):
        """Counter-based seed for the in-kernel dropout mask: device generator seed (no sync)
        mixed with this layer's call count.  Masks differ from torch's by construction."""
        # getattr defaults: modules un-pickled from a checkpoint written by the reference have neither field
        word = getattr(self, "_seed_word", None)
        if word is not None:
            # graph mode (sparch_amd.graph): the seed lives in device memory and is advanced by a captured op,
            # the kernels get its address (a kernel argument would be frozen at capture time)
            if not getattr(self, "_seed_word_owner_advances", False):  # (the captured step advances all layers' words at once)
                word.add_(1)
            return Fn.SEED_IN_MEMORY | word.data_ptr()
        self._calls = getattr(self, "_calls", 0) + 1
        index = getattr(self, "_layer_index", 0)
        base = torch.cuda.initial_seed() if device.type == "cuda" else torch.initial_seed()
        return (base * 0x9E3779B97F4A7C15 + (index + 1) * 0x100000001B3 + self._calls) & 0x7FFFFFFFFFFFFFFF

    def _cell_params(self):
        p = {"alpha": self.alpha}
        if self.kind in ("adLIF", "RadLIF"):
            p.update(beta=self.beta, a=self.a, b=self.b)
        if self.kind in ("RLIF", "RadLIF"):
            p["V"] = self.V.weight
        return p

    # ------------------------------------------------------------------ forward
    def forward_with_rate(self, x, states=None, states_ready=None, fp32_out=True):
        """Returns (spikes (B,T,H*(1+bidir)), firing_rate (H*(1+bidir),)).  Equivalent to the
        reference forward (e.g. snns.py:663-694) followed by `.mean(dim=(0,1))` (174).
        states: (u0, w0, s0) drawn ahead of time by SNN.forward (same generator, same order);
        states_ready: callable that makes the current stream wait for their upload, called after the
        projection GEMM has been enqueued (the upload runs under it);
        fp32_out=False: the caller feeds the result to another layer of this module only — the returned
        tensor is then a placeholder of the right shape whose VALUES ARE NOT WRITTEN (the spikes travel as the
        bf16 plane in its tag); SNN.forward uses it between its own layers."""
        Fn._require_device(x, "input")
        dirs = 2 if self.bidirectional else 1
        rows = x.shape[0] * dirs
        if self.batch_size != rows:
            self.batch_size = rows
        u0, w0, s0 = states if states is not None else self._draw_states(rows, x.device)
        p_drop = float(self.dropout) if self.training else 0.0
        is_bn = self.normalization == "batchnorm"
        in_scale, in_s16 = _spike_tag(x)
        cfg = {
            "kind": self.kind,
            "normalization": self.normalization if self.normalize else "none",
            "dirs": dirs,
            "training": bool(self.training),
            "theta": float(self.threshold),
            "p_drop": p_drop,
            "seed": self._dropout_seed(x.device) if p_drop > 0 else 0,
            "running_mean": self.norm.running_mean if is_bn else None,
            "running_var": self.norm.running_var if is_bn else None,
            # set when x came straight out of one of our spiking layers: entries are 0 or this constant
            "in_spike_scale": in_scale,
            "in_spike16": in_s16,  # the same spikes as a bf16 plane
            "states_ready": states_ready,
            "fp32_out": bool(fp32_out),
            # network input uploaded as bytes (functional.input_from_counts): its bf16 plane and the flag "exact"
            "in_plane": Fn.input_plane_of(x) if in_scale is None else None,
            # BatchNorm1d's counter: advanced by the statistics kernel (no separate launch; not on skipped steps)
            "num_batches_tracked": self.norm.num_batches_tracked if (is_bn and self.training) else None,
        }
        nw = self.norm.weight if self.normalize else None
        nb = self.norm.bias if self.normalize else None
        s, rate, s16 = Fn.SpikingLayerFn.apply(
            cfg, x, self.W.weight, self.W.bias, nw, nb, self.alpha,
            getattr(self, "beta", None), getattr(self, "a", None), getattr(self, "b", None),
            self.V.weight if hasattr(self, "V") else None, u0, w0, s0)
        # lets the next layer take the exact bf16-split GEMMs and read the spikes as a bf16 plane (half the bytes)
        has_plane = s16.numel() > 0
        _tag_spikes(s, 1.0 / (1.0 - p_drop), s16 if has_plane else None,
                    placeholder=has_plane and not fp32_out and not s.is_contiguous())
        return s, rate

    def forward(self, x):
        return self.forward_with_rate(x)[0]

    def _cell(self, Wx):
        """The cell alone on an already projected/normalised input (B',T,H) -> spikes (B',T,H),
        as the reference's `_lif_cell` / `_adlif_cell` / `_rlif_cell` / `_radlif_cell`."""
        Fn._require_device(Wx, "Wx")
        u0, w0, s0 = self._draw_states(Wx.shape[0], Wx.device)
        p = self._cell_params()
        return Fn.SpikingCellFn.apply(self.kind, float(self.threshold), Wx, p["alpha"], p.get("beta"),
                                      p.get("a"), p.get("b"), p.get("V"), u0, w0, s0, None)


class LIFLayer(_SpikingLayer):
    """Leaky integrate-and-fire layer without recurrence (reference LIFLayer, snns.py:179-303)."""
    kind = "LIF"

    def _lif_cell(self, Wx):
        return self._cell(Wx)


class adLIFLayer(_SpikingLayer):
    """Adaptive LIF layer without recurrence (reference adLIFLayer, snns.py:306-445)."""
    kind = "adLIF"

    def _adlif_cell(self, Wx):
        return self._cell(Wx)


class RLIFLayer(_SpikingLayer):
    """LIF layer with layer-wise recurrent weights V (reference RLIFLayer, snns.py:448-578)."""
    kind = "RLIF"

    def _rlif_cell(self, Wx):
        return self._cell(Wx)


class RadLIFLayer(_SpikingLayer):
    """Adaptive LIF layer with recurrent weights V (reference RadLIFLayer, snns.py:581-727)."""
    kind = "RadLIF"

    def _radlif_cell(self, Wx):
        return self._cell(Wx)


class ReadoutLayer(nn.Module):
    """Non-spiking leaky integrator whose output is the sum over time of softmax(u_t)
    (reference ReadoutLayer, snns.py:730-825).  No dropout is applied, as in the reference
    (the Dropout module exists at 791 but is never called)."""

    def __init__(self, input_size, hidden_size, batch_size, dropout=0.0, normalization="batchnorm",
                 use_bias=False):
        super().__init__()
        self.input_size = int(input_size)
        self.hidden_size = int(hidden_size)
        self.batch_size = batch_size
        self.dropout = dropout
        self.normalization = normalization
        self.use_bias = use_bias
        self.alpha_lim = [math.exp(-1 / 5), math.exp(-1 / 25)]

        self.W = nn.Linear(self.input_size, self.hidden_size, bias=use_bias)
        self.alpha = nn.Parameter(torch.empty(self.hidden_size))
        nn.init.uniform_(self.alpha, self.alpha_lim[0], self.alpha_lim[1])
        norm, self.normalize = _make_norm(normalization, self.hidden_size)
        if norm is not None:
            self.norm = norm
        self.drop = nn.Dropout(p=dropout)

    def forward(self, x, u0=None):
        Fn._require_device(x, "input")
        if u0 is None:
            u0 = _rand_to(x.shape[0], self.hidden_size, x.device)  # snns.py:812
        is_bn = self.normalization == "batchnorm"
        in_scale, in_s16 = _spike_tag(x)
        cfg = {
            "normalization": self.normalization if self.normalize else "none",
            "training": bool(self.training),
            "running_mean": self.norm.running_mean if is_bn else None,
            "running_var": self.norm.running_var if is_bn else None,
            "in_spike_scale": in_scale,
            "in_spike16": in_s16,  # the same spikes as a bf16 plane
            "num_batches_tracked": self.norm.num_batches_tracked if (is_bn and self.training) else None,
        }
        nw = self.norm.weight if self.normalize else None
        nb = self.norm.bias if self.normalize else None
        return Fn.ReadoutLayerFn.apply(cfg, x, self.W.weight, self.W.bias, nw, nb, self.alpha, u0)

    def _readout_cell(self, Wx):
        Fn._require_device(Wx, "Wx")
        u0 = _rand_to(Wx.shape[0], Wx.shape[2], Wx.device)
        return Fn.ReadoutCellFn.apply(Wx, self.alpha, u0)


_LAYER_CLASSES = {"LIF": LIFLayer, "adLIF": adLIFLayer, "RLIF": RLIFLayer, "RadLIF": RadLIFLayer}


def _global_forward_hooks():
    """True if torch.nn.modules.module has forward hooks registered for ALL modules (they would see a layer's output)."""
    import torch.nn.modules.module as _m

    return bool(getattr(_m, "_global_forward_hooks", None)) or bool(getattr(_m, "_global_forward_hooks_always_called", None))


class SNN(nn.Module):
    """Multi-layer spiking network (reference SNN, snns.py:39-176).

    forward(x: (batch, time, feat) or 4-D (batch, time, feat, channel)) ->
        (out, firing_rates): out is (batch, classes) with a readout layer, else
        (batch, time, feats); firing_rates has one entry per hidden neuron (all layers)."""

    def __init__(self, input_shape, layer_sizes, neuron_type="LIF", threshold=1.0, dropout=0.0,
                 normalization="batchnorm", use_bias=False, bidirectional=False, use_readout_layer=True):
        super().__init__()
        self.reshape = len(input_shape) > 3
        self.input_size = float(torch.prod(torch.tensor(input_shape[2:])))
        self.batch_size = input_shape[0]
        self.layer_sizes = layer_sizes
        self.num_layers = len(layer_sizes)
        self.num_outputs = layer_sizes[-1]
        self.neuron_type = neuron_type
        self.threshold = threshold
        self.dropout = dropout
        self.normalization = normalization
        self.use_bias = use_bias
        self.bidirectional = bidirectional
        self.use_readout_layer = use_readout_layer
        self.is_snn = True

        if neuron_type not in _SPIKING_KINDS:
            raise ValueError(f"Invalid neuron type {neuron_type}")

        self.snn = self._init_layers()

    def _init_layers(self):
        layers = nn.ModuleList([])
        layer_cls = _LAYER_CLASSES[self.neuron_type]
        n_hidden = self.num_layers - 1 if self.use_readout_layer else self.num_layers
        fan_in = self.input_size
        for i in range(n_hidden):
            layers.append(layer_cls(
                input_size=fan_in, hidden_size=self.layer_sizes[i], batch_size=self.batch_size,
                threshold=self.threshold, dropout=self.dropout, normalization=self.normalization,
                use_bias=self.use_bias, bidirectional=self.bidirectional))
            layers[-1]._layer_index = i
            fan_in = self.layer_sizes[i] * (1 + self.bidirectional)
        if self.use_readout_layer:
            layers.append(ReadoutLayer(
                input_size=fan_in, hidden_size=self.layer_sizes[-1], batch_size=self.batch_size,
                dropout=self.dropout, normalization=self.normalization, use_bias=self.use_bias))
        return layers

    def draw_states(self, batch, device, out=None):
        """Every layer's initial states for one forward, drawn from torch's global CPU generator in the
        reference's order (per hidden layer u, [w], s; then the readout's u): a list with one (u0, w0, s0)
        tuple per hidden layer and the readout's u0 tensor last.  out: the flat buffer of an earlier call
        (`.flat` of its result) to refill in place."""
        last = self.num_layers - 1
        shapes, plan = [], []
        for i, layer in enumerate(self.snn):
            if self.use_readout_layer and i == last:
                shapes.append((batch, layer.hidden_size))
                plan.append(None)
            else:
                rows = batch * (2 if layer.bidirectional else 1)
                adaptive = layer.kind in ("adLIF", "RadLIF")
                shapes += [(rows, layer.hidden_size)] * (3 if adaptive else 2)
                plan.append(adaptive)
        batch_list = _rand_batch(shapes, device, out=out)  # one staging buffer, one copy for the whole forward
        drawn = iter(batch_list)
        states = []
        for adaptive in plan:
            if adaptive is None:
                states.append(next(drawn))
            else:
                u0 = next(drawn)
                w0 = next(drawn) if adaptive else None
                states.append((u0, w0, next(drawn)))
        states = _StateList(states, batch_list.flat, batch_list.ready)
        return states

    def __getstate__(self):
        """Whole-module checkpoints (`torch.save(self.net)`, exp.py:462) hold the network, not a step's transient
        device state: the static states of a captured step (an event handle among them) stay out."""
        state = dict(self.__dict__)
        for k in ("_static_states", "_state_flat"):
            state.pop(k, None)
        return state

    def draw_states_into(self, static_states, batch):
        """The same draws (same generator, same order) written into the EXISTING device tensors `static_states`
        (an earlier draw_states result) — the static inputs of a captured step (sparch_amd.graph): one staging
        buffer and one stream-ordered copy into their common allocation."""
        flat = getattr(static_states, "flat", None)
        if flat is not None:
            self.draw_states(batch, flat.device, out=flat)
            return

        def fill(dst):
            dst.copy_(torch.rand(dst.shape[0], dst.shape[1], pin_memory=True), non_blocking=True)

        last = self.num_layers - 1
        for i, (layer, dst) in enumerate(zip(self.snn, static_states)):
            if self.use_readout_layer and i == last:
                fill(dst)
            else:
                u0, w0, s0 = dst
                assert u0.shape[0] == batch * (2 if layer.bidirectional else 1)
                fill(u0)
                if w0 is not None:
                    fill(w0)
                fill(s0)

    def forward(self, x):
        if self.reshape:
            if x.ndim == 4:
                x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3])
            else:
                raise NotImplementedError
        Fn._require_device(x, "input")
        # All initial states of this forward are drawn NOW, in the reference's order (layer by layer u, [w], s,
        # then the readout's u: snns.py:286-287 ... 812) from the same CPU generator, so the stream is the one
        # the reference consumes — but the host does its ~MBs of torch.rand while the GPU is still busy with
        # the previous step, instead of stalling the queue between two layers (120 us bubbles per layer in
        # the round-2 kernel trace).
        last = self.num_layers - 1
        states = getattr(self, "_static_states", None)  # graph mode: refilled by draw_states_into() per step
        if states is None:
            states = self.draw_states(x.shape[0], x.device)
        rates = []
        wait = getattr(states, "wait_ready", None)  # the states' upload runs under the first projection GEMM
        for i, layer in enumerate(self.snn):
            if self.use_readout_layer and i == last:
                if wait is not None:
                    wait()
                x = layer(x, u0=states[i])
            else:
                # between two layers of ours the spikes travel as a bf16 plane: no fp32 copy (unless somebody
                # else may look at the layer's output: forward hooks, the network's own output)
                inner = (i + 1 < len(self.snn) and Fn.USE_SPIKE_GEMM and Fn.USE_SPIKE16
                         and not layer._forward_hooks and not _global_forward_hooks())
                x, r = layer.forward_with_rate(x, states=states[i], states_ready=wait if i == 0 else None,
                                               fp32_out=not inner)
                if wait is not None:
                    wait()  # (no-op once the first layer's call has waited)
                rates.append(r)
        firing_rates = torch.cat(rates) if len(rates) > 1 else rates[0]
        return x, firing_rates
