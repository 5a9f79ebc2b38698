#!/usr/bin/env python3
"""
Generate tests/golden/*.npz by running the REAL reference implementation
(/root/reference/sparch/models/snns.py, imported read-only) on the CPU in the
build container.  The reference never travels to the GPU box; these fixtures
(inputs, parameters, random initial states, outputs, gradients) do.

    python tools/gen_golden.py            # rewrites tests/golden/

Determinism: every case seeds torch's global CPU generator, which is what the
reference draws its initial membrane / adaptation / spike states from
(snns.py:286-287, 423-425, 558-559, 700-702, 812).  The same seed is then
replayed to capture those states as explicit arrays (u0, w0, s0) so that the
oracle and the HIP path can be fed identical states.  pdrop = 0 everywhere:
dropout masks come from per-device generators and cannot match across devices.
"""
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from sparch.models import anns as ref_ann  # noqa: E402  (the reference's non-spiking baselines, f-4)
from sparch.models import snns as ref  # noqa: E402  (the reference)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
ADAPTIVE = {"LIF": False, "adLIF": True, "RLIF": False, "RadLIF": True}
RECURRENT = {"LIF": False, "adLIF": False, "RLIF": True, "RadLIF": True}


def npy(t):
    return t.detach().cpu().numpy().copy()


def draw_states(Bp, H, adaptive):
    st = {"u0": torch.rand(Bp, H)}
    if adaptive:
        st["w0"] = torch.rand(Bp, H)
    st["s0"] = torch.rand(Bp, H)
    return st


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ----------------------------------------------------------------------------
# 1. cells in isolation: Wx given, no projection / normalisation
# ----------------------------------------------------------------------------
def gen_cell(kind, B=4, T=24, H=32, seed=11):
    torch.manual_seed(seed)
    layer = getattr(ref, kind + "Layer")(
        input_size=8, hidden_size=H, batch_size=B, normalization="none"
    )
    # push a few raw parameters outside their clamp range (clamp gates their grad)
    with torch.no_grad():
        layer.alpha[0] = 0.5
        layer.alpha[1] = 0.99
        if ADAPTIVE[kind]:
            layer.beta[2] = 0.9
            layer.beta[3] = 0.999
            layer.a[4] = -1.5
            layer.a[5] = 1.25
            layer.b[6] = -0.1
            layer.b[7] = 2.5
    Wx = (torch.randn(B, T, H) * 1.5 + 0.3).requires_grad_(True)
    g_s = torch.randn(B, T, H)
    cell = getattr(layer, {"LIF": "_lif_cell", "adLIF": "_adlif_cell",
                           "RLIF": "_rlif_cell", "RadLIF": "_radlif_cell"}[kind])
    torch.manual_seed(seed + 1000)
    s = cell(Wx)
    torch.manual_seed(seed + 1000)
    st = draw_states(B, H, ADAPTIVE[kind])
    (s * g_s).sum().backward()
    arrays = dict(Wx=npy(Wx), g_s=npy(g_s), s=npy(s), dWx=npy(Wx.grad),
                  alpha=npy(layer.alpha), dalpha=npy(layer.alpha.grad),
                  **{k: npy(v) for k, v in st.items()})
    if ADAPTIVE[kind]:
        arrays.update(beta=npy(layer.beta), a=npy(layer.a), b=npy(layer.b),
                      dbeta=npy(layer.beta.grad), da=npy(layer.a.grad), db=npy(layer.b.grad))
    if RECURRENT[kind]:
        arrays.update(V=npy(layer.V.weight), dV=npy(layer.V.weight.grad))
    save(f"cell_{kind}", **arrays)


def gen_readout_cell(B=4, T=24, C=20, seed=21):
    torch.manual_seed(seed)
    layer = ref.ReadoutLayer(input_size=8, hidden_size=C, batch_size=B, normalization="none")
    with torch.no_grad():
        layer.alpha[0] = 0.5
        layer.alpha[1] = 0.99
    Wx = (torch.randn(B, T, C) * 2.0).requires_grad_(True)
    g = torch.randn(B, C)
    torch.manual_seed(seed + 1000)
    out = layer._readout_cell(Wx)
    torch.manual_seed(seed + 1000)
    u0 = torch.rand(B, C)
    (out * g).sum().backward()
    save("cell_readout", Wx=npy(Wx), g_out=npy(g), out=npy(out), dWx=npy(Wx.grad),
         alpha=npy(layer.alpha), dalpha=npy(layer.alpha.grad), u0=npy(u0))


# ----------------------------------------------------------------------------
# 2. whole SNN: projection + normalisation + cells + readout + loss
# ----------------------------------------------------------------------------
def gen_snn(name, neuron_type, layer_sizes, B, T, C, *, normalization="batchnorm",
            use_bias=False, bidirectional=False, use_readout_layer=True,
            use_regularizers=True, p_in=0.05, seed=1234, real_input=False):
    torch.manual_seed(seed)
    net = ref.SNN(input_shape=(B, None, C), layer_sizes=layer_sizes,
                  neuron_type=neuron_type, dropout=0.0, normalization=normalization,
                  use_bias=use_bias, bidirectional=bidirectional,
                  use_readout_layer=use_readout_layer)
    # de-trivialise affine norm parameters so their grads are exercised
    with torch.no_grad():
        for lay in net.snn:
            if hasattr(lay, "norm"):
                lay.norm.weight.uniform_(0.7, 1.3)
                lay.norm.bias.uniform_(-0.2, 0.2)
    gen = torch.Generator().manual_seed(4321)
    if real_input:
        x = torch.randn(B, T, C, generator=gen)
    else:
        x = (torch.rand(B, T, C, generator=gen) < p_in).float()
    n_cls = layer_sizes[-1]
    y = torch.randint(0, n_cls, (B,), generator=gen)

    params0 = {k: npy(v) for k, v in net.state_dict().items()}

    # ---- train-mode forward + backward (exp.py:359-376) ----
    net.train()
    fwd_seed = seed + 77
    torch.manual_seed(fwd_seed)
    out, rates = net(x)
    if use_readout_layer:
        loss = F.cross_entropy(out, y)
    else:
        loss = (out * out).mean()
    if use_regularizers:
        loss = loss + 0.5 * (F.relu(0.01 - rates).sum() + F.relu(rates - 0.5).sum())
    loss.backward()

    # replay the same seed to capture the initial states the reference drew
    torch.manual_seed(fwd_seed)
    Bp = B * (2 if bidirectional else 1)
    n_hidden = len(layer_sizes) - 1 if use_readout_layer else len(layer_sizes)
    states = {}
    for i in range(n_hidden):
        for k, v in draw_states(Bp, layer_sizes[i], ADAPTIVE[neuron_type]).items():
            states[f"init.{i}.{k}"] = npy(v)
    if use_readout_layer:
        states[f"init.{n_hidden}.u0"] = npy(torch.rand(B, n_cls))

    grads = {"grad." + k: npy(v.grad) for k, v in net.named_parameters()}
    stats1 = {"after." + k: npy(v) for k, v in net.state_dict().items() if "running" in k}

    # ---- eval-mode forward (exp.py:410-424) with the post-step running stats ----
    net.eval()
    with torch.no_grad():
        torch.manual_seed(fwd_seed)
        out_e, rates_e = net(x)

    x_store = npy(x).astype(np.float32) if real_input else npy(x).astype(np.uint8)
    save(name, x=x_store, y=npy(y), out=npy(out), rates=npy(rates), loss=npy(loss),
         out_eval=npy(out_e), rates_eval=npy(rates_e),
         cfg=np.array(json.dumps(dict(
             neuron_type=neuron_type, layer_sizes=layer_sizes, B=B, T=T, C=C,
             normalization=normalization, use_bias=use_bias, bidirectional=bidirectional,
             use_readout_layer=use_readout_layer, use_regularizers=use_regularizers,
             fwd_seed=fwd_seed, build_seed=seed))),
         **{"param." + k: v for k, v in params0.items()}, **states, **grads, **stats1)


# ----------------------------------------------------------------------------
# 2b. whole SNN on a fully DYADIC network: every matrix product is exact in fp32
# ----------------------------------------------------------------------------
class quantized_rand:
    """While active, torch.rand returns its usual draw rounded down to a multiple of 2^-4.  The reference's
    code is untouched: it still calls torch.rand for u0 / w0 / s0 (snns.py:286-287 ...), it just receives
    initial states on a coarse grid, so that the t = 0 product s0 @ V is exact like every later one."""

    def __enter__(self):
        self.orig = torch.rand
        orig = self.orig
        torch.rand = lambda *a, **k: torch.floor(orig(*a, **k) * 16.0) / 16.0
        return self

    def __exit__(self, *exc):
        torch.rand = self.orig


def gen_snn_dyadic(name, neuron_type, layer_sizes, B, T, C, *, normalization="none", use_bias=False,
                   bidirectional=False, p_in=0.3, w_gain=4.0, seed=2024, nonneg_a=False, grads_finite=True):
    """W, V (and biases) on a 2^-6 grid, 0/1 input, initial states on a 2^-4 grid: every W x, s @ V and
    s0 @ V partial sum is a small multiple of 2^-10 and therefore exact in fp32 in ANY summation order, so
    an implementation with a different GEMM k-order (MFMA vs the CPU sgemm) must still reproduce the
    reference's spikes bit for bit through all layers (normalization='none'); with batchnorm the statistics
    are order-dependent in the last bit, gamma is a power of two, `a` >= 0 keeps the subthreshold map
    contracting, and bit-equality is expected but not guaranteed by construction.
    Stored besides the usual outputs: every hidden layer's spike train (bit-packed)."""
    torch.manual_seed(seed)
    net = ref.SNN(input_shape=(B, None, C), layer_sizes=layer_sizes, neuron_type=neuron_type, dropout=0.0,
                  normalization=normalization, use_bias=use_bias, bidirectional=bidirectional,
                  use_readout_layer=True)
    q = lambda t, gain=1.0: torch.round(t * gain * 64.0) / 64.0  # noqa: E731
    with torch.no_grad():
        for lay in net.snn:
            lay.W.weight.copy_(q(lay.W.weight, w_gain))
            if use_bias:
                lay.W.bias.copy_(q(lay.W.bias, w_gain))
            if hasattr(lay, "V"):
                lay.V.weight.copy_(q(lay.V.weight))
            if nonneg_a and hasattr(lay, "a"):
                lay.a.abs_().mul_(0.01)
            if hasattr(lay, "norm"):
                lay.norm.weight.copy_(2.0 ** torch.randint(-1, 2, lay.norm.weight.shape).float())
                lay.norm.bias.copy_(q(torch.rand(lay.norm.bias.shape) * 1.0 + 1.0))
    gen = torch.Generator().manual_seed(4321)
    x = (torch.rand(B, T, C, generator=gen) < p_in).float()
    n_cls = layer_sizes[-1]
    y = torch.randint(0, n_cls, (B,), generator=gen)
    params0 = {k: npy(v) for k, v in net.state_dict().items()}

    spikes = {}
    hooks = [lay.register_forward_hook(lambda m, i, o, k=k: spikes.__setitem__(k, o.detach()))
             for k, lay in enumerate(list(net.snn)[:-1])]
    net.train()
    fwd_seed = seed + 77
    with quantized_rand():
        torch.manual_seed(fwd_seed)
        out, rates = net(x)
        loss = F.cross_entropy(out, y)
        loss.backward()
        torch.manual_seed(fwd_seed)
        Bp = B * (2 if bidirectional else 1)
        states = {}
        for i in range(len(layer_sizes) - 1):
            for k, v in draw_states(Bp, layer_sizes[i], ADAPTIVE[neuron_type]).items():
                states[f"init.{i}.{k}"] = npy(v)
        states[f"init.{len(layer_sizes) - 1}.u0"] = npy(torch.rand(B, n_cls))
    for h in hooks:
        h.remove()
    grads = {"grad." + k: npy(v.grad) for k, v in net.named_parameters()}
    n_bad = sum(int((~np.isfinite(g)).sum()) for g in grads.values())
    if grads_finite:
        assert n_bad == 0, f"{name}: {n_bad} non-finite gradient entries"
    stats1 = {"after." + k: npy(v) for k, v in net.state_dict().items() if "running" in k}
    packed = {}
    for k, s in spikes.items():
        s_np = npy(s)
        assert set(np.unique(s_np)) <= {0.0, 1.0}
        packed[f"spikes.{k}"] = np.packbits(s_np.astype(np.uint8), axis=None)
        packed[f"spikes.{k}.shape"] = np.array(s_np.shape)
        print(f"  layer {k}: firing rate {s_np.mean():.4f}")
    print(f"  non-finite gradient entries: {n_bad}")
    save(name, x=npy(x).astype(np.uint8), y=npy(y), out=npy(out), rates=npy(rates), loss=npy(loss),
         cfg=np.array(json.dumps(dict(
             neuron_type=neuron_type, layer_sizes=layer_sizes, B=B, T=T, C=C, normalization=normalization,
             use_bias=use_bias, bidirectional=bidirectional, use_readout_layer=True, use_regularizers=False,
             fwd_seed=fwd_seed, build_seed=seed, dyadic=True))),
         **{"param." + k: v for k, v in params0.items()}, **states, **grads, **stats1, **packed)


def gen_ann(name, ann_type, layer_sizes, B, T, C, *, normalization="batchnorm", use_bias=False,
            bidirectional=False, use_readout_layer=True, seed=4242):
    """Non-spiking baselines (anns.py): MLP / RNN / LiGRU / GRU + ReadoutLayerANN.  No random initial state
    (anns.py:331, 452, 584 start from zeros), dropout 0."""
    torch.manual_seed(seed)
    net = ref_ann.ANN(input_shape=(B, None, C), layer_sizes=layer_sizes, ann_type=ann_type, dropout=0.0,
                      normalization=normalization, use_bias=use_bias, bidirectional=bidirectional,
                      use_readout_layer=use_readout_layer)
    with torch.no_grad():
        for lay in net.ann:
            if hasattr(lay, "norm"):
                lay.norm.weight.uniform_(0.7, 1.3)
                lay.norm.bias.uniform_(-0.2, 0.2)
    gen = torch.Generator().manual_seed(999)
    x = torch.randn(B, T, C, generator=gen)
    n_cls = layer_sizes[-1]
    y = torch.randint(0, n_cls, (B,), generator=gen)
    params0 = {k: npy(v) for k, v in net.state_dict().items()}
    net.train()
    out, none = net(x)
    assert none is None
    loss = F.cross_entropy(out, y) if use_readout_layer else (out * out).mean()
    loss.backward()
    grads = {"grad." + k: npy(v.grad) for k, v in net.named_parameters()}
    stats1 = {"after." + k: npy(v) for k, v in net.state_dict().items() if "running" in k}
    net.eval()
    with torch.no_grad():
        out_e, _ = net(x)
    save(name, x=npy(x), y=npy(y), out=npy(out), loss=npy(loss), out_eval=npy(out_e),
         cfg=np.array(json.dumps(dict(ann_type=ann_type, layer_sizes=layer_sizes, B=B, T=T, C=C,
                                      normalization=normalization, use_bias=use_bias,
                                      bidirectional=bidirectional, use_readout_layer=use_readout_layer,
                                      build_seed=seed))),
         **{"param." + k: v for k, v in params0.items()}, **grads, **stats1)


def gen_reference_checkpoint():
    """A whole-module pickle exactly as the reference writes it (torch.save(self.net, ...), exp.py:462),
    plus the eval-mode output it produces, to test that such checkpoints load through this repo's
    `sparch.*` import shim and run on the HIP path."""
    torch.manual_seed(99)
    net = ref.SNN(input_shape=(4, None, 40), layer_sizes=[32, 32, 20], neuron_type="RadLIF", dropout=0.1,
                  normalization="batchnorm", use_bias=True, bidirectional=False)
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(4, 30, 40, generator=gen) < 0.2).float()
    net.train()
    torch.manual_seed(1)
    net(x)  # one training forward so the BatchNorm running statistics are non-trivial
    net.eval()
    with torch.no_grad():
        torch.manual_seed(2)
        out, rates = net(x)
    torch.save(net, os.path.join(OUT, "ref_checkpoint_RadLIF.pth"))
    save("ref_checkpoint_RadLIF_io", x=npy(x).astype(np.uint8), out_eval=npy(out), rates_eval=npy(rates),
         **{"param." + k: npy(v) for k, v in net.state_dict().items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    for kind in ("LIF", "adLIF", "RLIF", "RadLIF"):
        gen_cell(kind)
    gen_readout_cell()
    # BASELINE.json configs[0]: LIF [128,128,20], B=4, T=100, C=700 (SHD shape)
    gen_snn("snn_cfg1_LIF", "LIF", [128, 128, 20], 4, 100, 700)
    # small instances of every other neuron type / option on the hot path
    gen_snn("snn_adLIF_bn", "adLIF", [48, 48, 20], 6, 40, 64, p_in=0.15)
    gen_snn("snn_RLIF_nonorm_bias", "RLIF", [32, 32, 20], 5, 30, 40,
            normalization="none", use_bias=True, p_in=0.3, use_regularizers=False)
    gen_snn("snn_RadLIF_bn", "RadLIF", [64, 64, 35], 8, 50, 96, p_in=0.1)
    gen_snn("snn_RadLIF_bidir_bn", "RadLIF", [32, 32, 35], 4, 30, 40,
            bidirectional=True, p_in=0.15)
    gen_snn("snn_adLIF_layernorm", "adLIF", [32, 32, 20], 4, 25, 40,
            normalization="layernorm", real_input=True)
    gen_snn("snn_LIF_noreadout", "LIF", [32, 24], 4, 20, 40,
            use_readout_layer=False, use_regularizers=False, p_in=0.2)
    gen_reference_checkpoint()
    # fully dyadic networks: spikes reproducible bit for bit by any summation order (see gen_snn_dyadic)
    gen_snn_dyadic("dyadic_RadLIF_none", "RadLIF", [64, 64, 20], 8, 40, 64)
    gen_snn_dyadic("dyadic_RLIF_none_bias", "RLIF", [64, 32, 20], 6, 32, 48, use_bias=True, seed=2025, w_gain=10.0,
                   p_in=0.4)
    gen_snn_dyadic("dyadic_RadLIF_bidir_none", "RadLIF", [32, 32, 20], 4, 24, 40, bidirectional=True, seed=2026)
    gen_snn_dyadic("dyadic_RadLIF_bn", "RadLIF", [64, 64, 20], 8, 32, 64, normalization="batchnorm",
                   nonneg_a=True, w_gain=1.0, seed=2027)
    # long sequence (BASELINE configs[4] is T = 1000): neurons whose (u, w) map is unstable overflow fp32 and
    # the reference's own alpha / beta / a gradients become non-finite; the fixture pins WHICH entries
    gen_snn_dyadic("dyadic_RadLIF_T1000", "RadLIF", [64, 64, 20], 4, 1000, 32, seed=2028, grads_finite=False)
    # f-4: non-spiking baselines
    gen_ann("ann_MLP_bn", "MLP", [48, 48, 20], 6, 30, 40)
    gen_ann("ann_MLP_ln_bias_noreadout", "MLP", [32, 24], 4, 20, 40, normalization="layernorm", use_bias=True,
            use_readout_layer=False)
    gen_ann("ann_RNN_bn", "RNN", [48, 48, 20], 6, 30, 40)
    gen_ann("ann_RNN_bidir", "RNN", [32, 32, 20], 4, 24, 40, bidirectional=True, normalization="none")
    gen_ann("ann_LiGRU_bn", "LiGRU", [32, 32, 20], 4, 24, 40)
    gen_ann("ann_GRU_bn", "GRU", [32, 32, 20], 4, 24, 40)


if __name__ == "__main__":
    main()
