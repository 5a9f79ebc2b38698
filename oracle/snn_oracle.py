"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (plain eager PyTorch, fp32) of the reference's recurrent
spiking-layer training path.  Only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import this file; the product package
`sparch_amd` never does (it fails loudly when the HIP library is missing).

Parity status: PINNED.  tools/gen_golden.py imports the real reference
(/root/reference/sparch/models/snns.py) in the build container and writes
tests/golden/*.npz; tests/test_oracle_golden.py requires this restatement to
reproduce those fixtures (spikes bit-equal, everything else to fp32 rounding).

The restatement is functional (parameters come in as a dict keyed like the
reference's state_dict) and takes the random initial states as explicit
inputs, so that a HIP run and an oracle run can share them.  Every function
cites the reference lines it follows.
"""
import math

import torch
import torch.nn.functional as F

# Clamp ranges of the neuron parameters.
#   alpha: snns.py:229 / 356 / 498 / 631 / 774     beta: snns.py:357 / 632
#   a:     snns.py:358 / 633                        b:    snns.py:359 / 634
ALPHA_LIM = (math.exp(-1 / 5), math.exp(-1 / 25))
BETA_LIM = (math.exp(-1 / 30), math.exp(-1 / 120))
A_LIM = (-1.0, 1.0)
B_LIM = (0.0, 2.0)

BN_MOMENTUM = 0.05  # snns.py:240
BN_EPS = 1e-5  # nn.BatchNorm1d / nn.LayerNorm default, snns.py:240, 243

ADAPTIVE = {"LIF": False, "adLIF": True, "RLIF": False, "RadLIF": True}
RECURRENT = {"LIF": False, "adLIF": False, "RLIF": True, "RadLIF": True}


class _Boxcar(torch.autograd.Function):
    """Heaviside forward, box-car surrogate backward (snns.py:26-36).

    forward:  s = 1[x > 0]                      (strict >, snns.py:29)
    backward: g_x = g_s with the entries where x <= -0.5 or x > 0.5 SET to zero (snns.py:33-35).
    Assignment, not multiplication: an infinite / NaN upstream gradient is zeroed outside the box-car, and a
    NaN x (both comparisons false) lets the gradient through — visible in the long-sequence fixture, where
    unstable neurons overflow fp32.
    """

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return (x > 0).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g_x = g.clone()
        g_x[x <= -0.5] = 0
        g_x[x > 0.5] = 0
        return g_x


def spike(x):
    return _Boxcar.apply(x)


def spiking_cell(kind, Wx, p, u0, w0, s0, theta=1.0):
    """One spiking cell over the whole sequence.

    kind in {LIF, adLIF, RLIF, RadLIF}; Wx (B', T, H) post-normalisation;
    p holds 'alpha' [, 'beta', 'a', 'b'] [, 'V' (H, H) = V.weight];
    u0, w0, s0 (B', H) are the random initial states (w0 unused unless adaptive).

    Follows _lif_cell snns.py:282-303, _adlif_cell 419-445, _rlif_cell 554-578,
    _radlif_cell 696-727, keeping their operation order so that CPU results are
    bit-identical:
        w_t = beta*w + a*u + b*s                    (438 / 718, previous u and s)
        u_t = alpha*(u - s) + (1-alpha)*(Wx_t [+ s@V] [- w_t])   (297/439/572/719-721)
        s_t = H(u_t - theta)                        (300 / 442 / 575 / 724)
    `s @ V` uses V.weight un-transposed with its diagonal zeroed (566 / 712).
    """
    adaptive, recurrent = ADAPTIVE[kind], RECURRENT[kind]
    alpha = torch.clamp(p["alpha"], min=ALPHA_LIM[0], max=ALPHA_LIM[1])
    if adaptive:
        beta = torch.clamp(p["beta"], min=BETA_LIM[0], max=BETA_LIM[1])
        a = torch.clamp(p["a"], min=A_LIM[0], max=A_LIM[1])
        b = torch.clamp(p["b"], min=B_LIM[0], max=B_LIM[1])
    if recurrent:
        Vm = p["V"].clone().fill_diagonal_(0)

    u, s, w = u0, s0, w0
    out = []
    for t in range(Wx.shape[1]):
        drive = Wx[:, t, :]
        if adaptive:
            w = beta * w + a * u + b * s
        if recurrent:
            drive = drive + torch.matmul(s, Vm)
        if adaptive:
            drive = drive - w
        u = alpha * (u - s) + (1 - alpha) * drive
        s = spike(u - theta)
        out.append(s)
    return torch.stack(out, dim=1)


def readout_cell(Wx, alpha_raw, u0):
    """Non-spiking leaky integrator with a softmax-sum output (snns.py:808-825):
    u_t = alpha*u + (1-alpha)*Wx_t (822);  out += softmax(u_t, dim=1) (823)."""
    alpha = torch.clamp(alpha_raw, min=ALPHA_LIM[0], max=ALPHA_LIM[1])
    u = u0
    out = torch.zeros(Wx.shape[0], Wx.shape[2], dtype=Wx.dtype)
    for t in range(Wx.shape[1]):
        u = alpha * u + (1 - alpha) * Wx[:, t, :]
        out = out + F.softmax(u, dim=1)
    return out


def _normalise(Wx, p, prefix, normalization, training, stats):
    """Normalisation on the (B'*T, H) view (snns.py:264-266): BatchNorm1d with
    momentum 0.05 (240) or LayerNorm (243); any other string means none (238-244).
    `stats` (a dict) receives updated running statistics under the state_dict
    names, the way nn.BatchNorm1d would mutate its buffers in train mode."""
    if normalization not in ("batchnorm", "layernorm"):
        return Wx
    Bp, T, H = Wx.shape
    flat = Wx.reshape(Bp * T, H)
    if normalization == "batchnorm":
        rm = p[prefix + "norm.running_mean"].clone()
        rv = p[prefix + "norm.running_var"].clone()
        flat = F.batch_norm(
            flat,
            rm,
            rv,
            p[prefix + "norm.weight"],
            p[prefix + "norm.bias"],
            training,
            BN_MOMENTUM,
            BN_EPS,
        )
        if stats is not None:
            stats[prefix + "norm.running_mean"] = rm
            stats[prefix + "norm.running_var"] = rv
    else:
        flat = F.layer_norm(
            flat, (H,), p[prefix + "norm.weight"], p[prefix + "norm.bias"], BN_EPS
        )
    return flat.reshape(Bp, T, H)


def hidden_layer(kind, x, p, prefix, init, *, normalization="batchnorm",
                 bidirectional=False, training=True, theta=1.0, stats=None,
                 drop_mask=None):
    """{LIF,adLIF,RLIF,RadLIF}Layer.forward (snns.py:249-280, 386-417, 521-552,
    663-694).  `init` = dict(u0, s0[, w0]) of shape (B', H) with B' = B*(1+bidir).
    `drop_mask` (B, T, H*(1+bidir)), already scaled by 1/(1-p), stands in for
    nn.Dropout (278): device RNG streams cannot match, so parity runs use p=0
    and pass None."""
    if bidirectional:  # 252-254: time-flipped copy stacked on the batch axis
        x = torch.cat([x, x.flip(1)], dim=0)
    Wx = F.linear(x, p[prefix + "W.weight"], p.get(prefix + "W.bias"))  # 261
    Wx = _normalise(Wx, p, prefix, normalization, training, stats)  # 264-266
    cell_p = {"alpha": p[prefix + "alpha"]}
    if ADAPTIVE[kind]:
        cell_p.update(beta=p[prefix + "beta"], a=p[prefix + "a"], b=p[prefix + "b"])
    if RECURRENT[kind]:
        cell_p["V"] = p[prefix + "V.weight"]
    s = spiking_cell(kind, Wx, cell_p, init["u0"], init.get("w0"), init["s0"], theta)
    if bidirectional:  # 272-275: un-flip the backward half, stack on features
        s_f, s_b = s.chunk(2, dim=0)
        s = torch.cat([s_f, s_b.flip(1)], dim=2)
    if drop_mask is not None:  # 278
        s = s * drop_mask
    return s


def readout_layer(x, p, prefix, u0, *, normalization="batchnorm", training=True,
                  stats=None):
    """ReadoutLayer.forward (snns.py:793-806).  No dropout is applied (the
    module builds one at 791 and never calls it)."""
    Wx = F.linear(x, p[prefix + "W.weight"], p.get(prefix + "W.bias"))  # 796
    Wx = _normalise(Wx, p, prefix, normalization, training, stats)  # 799-801
    return readout_cell(Wx, p[prefix + "alpha"], u0)  # 804


def snn_forward(x, p, *, neuron_type, num_layers, init_states, normalization="batchnorm",
                bidirectional=False, use_readout_layer=True, training=True,
                theta=1.0, stats=None, drop_masks=None, spikes_out=None):
    """SNN.forward (snns.py:157-176): layer loop, firing rates = mean over
    (batch, time) of the concatenated post-dropout hidden outputs (171-174).
    `p` is keyed like the reference state_dict ('snn.{i}.alpha', ...);
    `init_states[i]` is dict(u0, s0[, w0]) for hidden layer i and dict(u0) for
    the readout.  `spikes_out` (a list) receives every hidden layer's output."""
    if x.ndim == 4:  # 160-162
        x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3])
    elif x.ndim != 3:
        raise NotImplementedError  # 164
    n_hidden = num_layers - 1 if use_readout_layer else num_layers
    spikes = []
    for i in range(n_hidden):
        x = hidden_layer(
            neuron_type, x, p, f"snn.{i}.", init_states[i],
            normalization=normalization, bidirectional=bidirectional,
            training=training, theta=theta, stats=stats,
            drop_mask=None if drop_masks is None else drop_masks[i],
        )
        spikes.append(x)
    if spikes_out is not None:
        spikes_out.extend(spikes)
    rates = torch.cat(spikes, dim=2).mean(dim=(0, 1))  # 174
    if use_readout_layer:
        x = readout_layer(
            x, p, f"snn.{n_hidden}.", init_states[n_hidden]["u0"],
            normalization=normalization, training=training, stats=stats,
        )
    return x, rates


def draw_init_states(batch, layer_sizes, neuron_type, bidirectional=False,
                     use_readout_layer=True):
    """Draw the random initial states from torch's global CPU generator in the
    reference's order: per hidden layer u, [w], s of shape (B', H)
    (snns.py:286-287, 423-425, 558-559, 700-702); readout u of shape (B, classes)
    (812).  Call right after torch.manual_seed(seed) to reproduce what a
    reference forward would have drawn with that seed."""
    n_hidden = len(layer_sizes) - 1 if use_readout_layer else len(layer_sizes)
    Bp = batch * (2 if bidirectional else 1)
    states = []
    for i in range(n_hidden):
        st = {"u0": torch.rand(Bp, layer_sizes[i])}
        if ADAPTIVE[neuron_type]:
            st["w0"] = torch.rand(Bp, layer_sizes[i])
        st["s0"] = torch.rand(Bp, layer_sizes[i])
        states.append(st)
    if use_readout_layer:
        states.append({"u0": torch.rand(batch, layer_sizes[-1])})
    return states


def train_step_loss(out, rates, y, *, use_regularizers=False, reg_factor=0.5,
                    reg_fmin=0.01, reg_fmax=0.5):
    """Loss of the train step body (exp.py:362, 369-372): cross-entropy applied to
    the softmax-sum output, plus the optional firing-rate regulariser."""
    loss = F.cross_entropy(out, y)
    if use_regularizers:
        quiet = F.relu(reg_fmin - rates).sum()
        burst = F.relu(rates - reg_fmax).sum()
        loss = loss + reg_factor * (quiet + burst)
    return loss
