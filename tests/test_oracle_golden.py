"""
Pins the oracle (oracle/snn_oracle.py, oracle/bptt_numpy.py) against golden vectors
produced by the real reference (tools/gen_golden.py).  CPU only.

Bar: spikes / softmax-sum outputs bit-equal (same op order on the same CPU);
autograd grads of the torch restatement bit-equal or to fp32 rounding; the
hand-derived BPTT recurrences (what the HIP backward kernels implement) to a
stated fp32 tolerance.
"""
import numpy as np
import pytest
import torch

from oracle import bptt_numpy as bp
from oracle import snn_oracle as orc
from tests.golden_io import CELL_KINDS, DYADIC_CASES, DYADIC_LONG, SNN_CASES, layer_spikes, load, snn_case

torch.set_num_threads(1)


def _cell_params(z, kind, grad=False):
    keys = ["alpha"] + (["beta", "a", "b"] if orc.ADAPTIVE[kind] else []) + (["V"] if orc.RECURRENT[kind] else [])
    return {k: torch.from_numpy(z[k]).clone().requires_grad_(grad) for k in keys}


@pytest.mark.parametrize("kind", CELL_KINDS)
def test_cell_torch_restatement_matches_reference(kind):
    z = load(f"cell_{kind}")
    p = _cell_params(z, kind, grad=True)
    Wx = torch.from_numpy(z["Wx"]).requires_grad_(True)
    w0 = torch.from_numpy(z["w0"]) if "w0" in z else None
    s = orc.spiking_cell(kind, Wx, p, torch.from_numpy(z["u0"]), w0, torch.from_numpy(z["s0"]))
    assert np.array_equal(s.detach().numpy(), z["s"]), "spikes must be bit-identical"
    (s * torch.from_numpy(z["g_s"])).sum().backward()
    np.testing.assert_allclose(Wx.grad.numpy(), z["dWx"], rtol=1e-6, atol=1e-7)
    for k in p:
        np.testing.assert_allclose(p[k].grad.numpy(), z["d" + k], rtol=1e-5, atol=1e-6)
    if orc.RECURRENT[kind]:
        assert np.all(np.diag(p["V"].grad.numpy()) == 0)


@pytest.mark.parametrize("kind", CELL_KINDS)
def test_cell_manual_bptt_matches_reference_autograd(kind):
    """The reverse-time recurrences the HIP kernels implement == reference autograd."""
    z = load(f"cell_{kind}")
    p = {k: z[k] for k in ("alpha", "beta", "a", "b", "V") if k in z}
    w0 = z.get("w0")
    S, U, W = bp.cell_forward(kind, z["Wx"], p, z["u0"], w0, z["s0"])
    assert np.array_equal(S, z["s"]), "numpy forward spikes must be bit-identical"
    g = bp.cell_backward(kind, z["g_s"], z["Wx"], p, z["u0"], w0, z["s0"], U, W)
    np.testing.assert_allclose(g["dWx"], z["dWx"], rtol=2e-5, atol=2e-6)
    scale = {k: max(1.0, float(np.abs(z[k]).max())) for k in g}
    for k in g:
        if k == "dWx":
            continue
        assert np.abs(g[k] - z[k]).max() <= 5e-5 * scale[k], k
    # clamp gating: raw parameters outside [min, max] get exactly zero gradient
    assert g["dalpha"][0] == 0 and g["dalpha"][1] == 0
    if orc.ADAPTIVE[kind]:
        assert g["dbeta"][2] == 0 and g["dbeta"][3] == 0
        assert g["da"][4] == 0 and g["da"][5] == 0
        assert g["db"][6] == 0 and g["db"][7] == 0
    if orc.RECURRENT[kind]:
        assert np.all(np.diag(g["dV"]) == 0)


def test_readout_cell_matches_reference():
    z = load("cell_readout")
    Wx = torch.from_numpy(z["Wx"]).requires_grad_(True)
    alpha = torch.from_numpy(z["alpha"]).requires_grad_(True)
    out = orc.readout_cell(Wx, alpha, torch.from_numpy(z["u0"]))
    assert np.array_equal(out.detach().numpy(), z["out"])
    (out * torch.from_numpy(z["g_out"])).sum().backward()
    np.testing.assert_allclose(Wx.grad.numpy(), z["dWx"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(alpha.grad.numpy(), z["dalpha"], rtol=1e-5, atol=1e-6)
    # manual recurrences
    o2, U = bp.readout_forward(z["Wx"], z["alpha"], z["u0"])
    np.testing.assert_allclose(o2, z["out"], rtol=1e-6, atol=1e-6)
    g = bp.readout_backward(z["g_out"], z["Wx"], z["alpha"], z["u0"], U)
    np.testing.assert_allclose(g["dWx"], z["dWx"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(g["dalpha"], z["dalpha"], rtol=1e-4, atol=1e-5)
    assert g["dalpha"][0] == 0 and g["dalpha"][1] == 0


@pytest.mark.parametrize("name", SNN_CASES)
def test_snn_oracle_matches_reference(name):
    cfg, x, y, params, init, z = snn_case(name)
    p = {k: v.clone().requires_grad_(v.dtype == torch.float32 and "running" not in k) for k, v in params.items()}
    stats = {}
    out, rates = orc.snn_forward(
        x, p, neuron_type=cfg["neuron_type"], num_layers=len(cfg["layer_sizes"]),
        init_states=init, normalization=cfg["normalization"], bidirectional=cfg["bidirectional"],
        use_readout_layer=cfg["use_readout_layer"], training=True, stats=stats)
    assert np.array_equal(out.detach().numpy(), z["out"]), "train-mode output must be bit-identical"
    assert np.array_equal(rates.detach().numpy(), z["rates"])
    if cfg["use_readout_layer"]:
        loss = orc.train_step_loss(out, rates, y, use_regularizers=cfg["use_regularizers"])
    else:
        loss = (out * out).mean()
    assert np.array_equal(loss.detach().numpy(), z["loss"])
    loss.backward()
    for k, v in p.items():
        if v.requires_grad:
            np.testing.assert_allclose(v.grad.numpy(), z["grad." + k], rtol=2e-5, atol=1e-6, err_msg=k)
    for k, v in stats.items():
        np.testing.assert_allclose(v.numpy(), z["after." + k], rtol=1e-6, atol=1e-8, err_msg=k)
    # eval mode with the updated running statistics
    p_eval = {k: v.detach() for k, v in p.items()}
    p_eval.update({k: v for k, v in stats.items()})
    with torch.no_grad():
        out_e, rates_e = orc.snn_forward(
            x, p_eval, neuron_type=cfg["neuron_type"], num_layers=len(cfg["layer_sizes"]),
            init_states=init, normalization=cfg["normalization"], bidirectional=cfg["bidirectional"],
            use_readout_layer=cfg["use_readout_layer"], training=False)
    assert np.array_equal(out_e.numpy(), z["out_eval"])
    assert np.array_equal(rates_e.numpy(), z["rates_eval"])


@pytest.mark.parametrize("name", DYADIC_CASES + [DYADIC_LONG])
def test_snn_oracle_matches_reference_on_dyadic_networks(name):
    """Whole networks with dyadic W / V / initial states (every product exact): per-layer spikes, output, loss
    and every gradient against the reference; for the T = 1000 case also WHICH gradient entries the reference
    itself leaves non-finite (unstable neurons overflow fp32)."""
    cfg, x, y, params, init, z = snn_case(name)
    p = {k: v.clone().requires_grad_(v.dtype == torch.float32 and "running" not in k) for k, v in params.items()}
    spikes = []
    out, rates = orc.snn_forward(
        x, p, neuron_type=cfg["neuron_type"], num_layers=len(cfg["layer_sizes"]), init_states=init,
        normalization=cfg["normalization"], bidirectional=cfg["bidirectional"], training=True, stats={},
        spikes_out=spikes)
    for k, s in enumerate(spikes):
        assert np.array_equal(s.detach().numpy(), layer_spikes(z, k)), f"layer {k} spikes"
    assert np.array_equal(out.detach().numpy(), z["out"])
    loss = orc.train_step_loss(out, rates, y)
    loss.backward()
    n_bad = 0
    for k, v in p.items():
        if not v.requires_grad:
            continue
        g, g_ref = v.grad.numpy(), z["grad." + k]
        assert np.array_equal(np.isfinite(g), np.isfinite(g_ref)), f"non-finite pattern of grad {k}"
        ok = np.isfinite(g_ref)
        n_bad += int((~ok).sum())
        np.testing.assert_allclose(g[ok], g_ref[ok], rtol=2e-5, atol=1e-6, err_msg=k)
    assert (n_bad > 0) == (name == DYADIC_LONG)


def test_draw_init_states_order_matches_reference_rng():
    """torch.manual_seed(s) + draw_init_states == what the reference drew (captured in the fixture)."""
    cfg, x, y, params, init, z = snn_case("snn_RadLIF_bidir_bn")
    torch.manual_seed(cfg["fwd_seed"])
    st = orc.draw_init_states(cfg["B"], cfg["layer_sizes"], cfg["neuron_type"], cfg["bidirectional"])
    for i, d in enumerate(st):
        for k, v in d.items():
            assert np.array_equal(v.numpy(), z[f"init.{i}.{k}"]), (i, k)


# ------------------------------------------------------------------ f-4: non-spiking baselines (anns.py)
ANN_CASES = ["ann_MLP_bn", "ann_MLP_ln_bias_noreadout", "ann_RNN_bn", "ann_RNN_bidir", "ann_LiGRU_bn", "ann_GRU_bn"]


@pytest.mark.parametrize("name", ANN_CASES)
def test_ann_oracle_matches_reference_fixture(name):
    """oracle/ann_oracle.py against outputs, loss, gradients, running statistics and the eval-mode output of
    the real reference's ANN (tools/gen_golden.py)."""
    import json

    import torch.nn.functional as F

    from oracle import ann_oracle as ao

    z = load(name)
    cfg = json.loads(str(z["cfg"]))
    p = {k[len("param."):]: torch.tensor(z[k]).requires_grad_(z[k].dtype.kind == "f" and "running" not in k)
         for k in z if k.startswith("param.") and "num_batches" not in k}
    x, y = torch.tensor(z["x"]), torch.tensor(z["y"])
    running = {k: v.detach().clone() for k, v in p.items() if "running" in k}
    out = ao.ann_forward(cfg, p, x, training=True, running=running)
    np.testing.assert_allclose(out.detach().numpy(), z["out"], rtol=2e-5, atol=2e-6)
    loss = F.cross_entropy(out, y) if cfg["use_readout_layer"] else (out * out).mean()
    np.testing.assert_allclose(float(loss.detach()), float(z["loss"]), rtol=1e-5)
    loss.backward()
    for k in z:
        if k.startswith("grad."):
            g = p[k[len("grad."):]].grad
            np.testing.assert_allclose(g.numpy(), z[k], rtol=2e-4, atol=2e-6, err_msg=k)
        if k.startswith("after."):
            np.testing.assert_allclose(running[k[len("after."):]].numpy(), z[k], rtol=1e-5, atol=1e-6, err_msg=k)
    with torch.no_grad():
        out_e = ao.ann_forward(cfg, p, x, training=False, running=running)
    np.testing.assert_allclose(out_e.numpy(), z["out_eval"], rtol=2e-5, atol=2e-6)
