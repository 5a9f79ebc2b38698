"""CPU: host-side logic of the drop-in mirror — construction parity with the reference
(fixtures from tools/gen_golden.py), state_dict surface, error behaviour, CLI flags."""
import numpy as np
import pytest
import torch

import sparch_amd
from tests.golden_io import snn_case


@pytest.mark.parametrize("name", ["snn_RadLIF_bidir_bn", "snn_adLIF_bn", "snn_RLIF_nonorm_bias", "snn_cfg1_LIF",
                                  "snn_adLIF_layernorm", "snn_LIF_noreadout"])
def test_construction_matches_reference_rng_stream(name):
    """torch.manual_seed(s); SNN(...) reproduces the reference's initial parameters (construction-time
    RNG draw order, snns.py:233-235, 363-372, 502-507, 638-649) and its state_dict keys/shapes."""
    cfg, x, y, params, init, z = snn_case(name)
    torch.manual_seed(cfg["build_seed"])
    net = sparch_amd.SNN((cfg["B"], None, cfg["C"]), cfg["layer_sizes"], neuron_type=cfg["neuron_type"],
                         normalization=cfg["normalization"], use_bias=cfg["use_bias"],
                         bidirectional=cfg["bidirectional"], use_readout_layer=cfg["use_readout_layer"])
    sd = net.state_dict()
    assert list(sd.keys()) == list(params.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(params[k].shape), k
        if "norm.weight" in k or "norm.bias" in k:
            continue  # the fixture generator re-randomised the affine norm parameters afterwards
        if k.endswith("V.weight"):  # orthogonal init runs a QR: LAPACK rounding may differ between hosts
            np.testing.assert_allclose(v.numpy(), z["param." + k], rtol=0, atol=2e-6, err_msg=k)
        else:
            assert np.array_equal(v.numpy(), z["param." + k]), k


def test_api_surface_and_errors():
    with pytest.raises(ValueError, match="Invalid neuron type"):
        sparch_amd.SNN((4, None, 8), [8, 4], neuron_type="GRU")
    net = sparch_amd.SNN((4, None, 8, 2), [8, 4], neuron_type="LIF")
    assert net.is_snn and net.reshape and net.num_layers == 2 and net.input_size == 16.0
    lay = sparch_amd.RadLIFLayer(input_size=8, hidden_size=16, batch_size=4, bidirectional=True)
    assert lay.batch_size == 8 and lay.V.weight.shape == (16, 16)
    assert set(dict(lay.named_parameters())) == {"alpha", "beta", "a", "b", "W.weight", "V.weight",
                                                 "norm.weight", "norm.bias"}
    for meth in ("_lif_cell", "_adlif_cell", "_rlif_cell", "_radlif_cell"):
        assert sum(hasattr(c, meth) for c in (sparch_amd.LIFLayer, sparch_amd.adLIFLayer,
                                              sparch_amd.RLIFLayer, sparch_amd.RadLIFLayer)) == 1
    assert hasattr(sparch_amd.ReadoutLayer, "_readout_cell")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(4, 3, 8, 2))


def test_reference_import_path_shim():
    from sparch.models.snns import SNN, LIFLayer, RadLIFLayer, ReadoutLayer, SpikeFunctionBoxcar  # noqa: F401
    assert SNN is sparch_amd.SNN


def test_spike_function_boxcar_semantics():
    x = torch.tensor([-0.6, -0.5, -0.25, 0.0, 0.25, 0.5, 0.75], requires_grad=True)
    s = sparch_amd.SpikeFunctionBoxcar.apply(x)
    assert s.tolist() == [0, 0, 0, 0, 1, 1, 1]            # strict > 0 (snns.py:29)
    s.sum().backward()
    assert x.grad.tolist() == [0, 0, 1, 1, 1, 1, 0]        # (-0.5, 0.5] (snns.py:34-35)


def test_reference_written_checkpoint_loads_through_shim():
    """Whole-module pickle written by the REAL reference (torch.save(self.net), exp.py:462; fixture made by
    tools/gen_golden.py): un-pickles into this repo's classes via the `sparch.*` shim with identical weights.
    weights_only=False is required for module pickles; the file is produced by our own generator script."""
    import os
    from tests.golden_io import GOLDEN, load
    net = torch.load(os.path.join(GOLDEN, "ref_checkpoint_RadLIF.pth"), weights_only=False)
    assert isinstance(net, sparch_amd.SNN) and isinstance(net.snn[0], sparch_amd.RadLIFLayer)
    assert isinstance(net.snn[2], sparch_amd.ReadoutLayer) and net.is_snn and not net.training
    z = load("ref_checkpoint_RadLIF_io")
    for k, v in net.state_dict().items():
        assert np.array_equal(v.numpy(), z["param." + k]), k
    assert net.snn[0].W.bias is not None and net.snn[0].dropout == 0.1


# ------------------------------------------------------------------ f-4: non-spiking baselines
@pytest.mark.parametrize("name", ["ann_MLP_bn", "ann_RNN_bidir", "ann_LiGRU_bn", "ann_GRU_bn"])
def test_ann_construction_matches_reference_rng_and_keys(name):
    """Same seed -> same state_dict keys, shapes and initial values as the reference's ANN (fixtures hold the
    reference's freshly constructed parameters, with the norm affine parameters re-drawn afterwards).
    Orthogonal init goes through QR, whose LAPACK result can differ in the last bits between hosts: V
    matrices are compared at 2e-6."""
    import json

    from sparch_amd.anns import ANN
    from tests.golden_io import load

    z = load(name)
    cfg = json.loads(str(z["cfg"]))
    torch.manual_seed(cfg["build_seed"])
    net = ANN(input_shape=(cfg["B"], None, cfg["C"]), layer_sizes=cfg["layer_sizes"], ann_type=cfg["ann_type"],
              dropout=0.0, normalization=cfg["normalization"], use_bias=cfg["use_bias"],
              bidirectional=cfg["bidirectional"], use_readout_layer=cfg["use_readout_layer"])
    sd = net.state_dict()
    ref_keys = sorted(k[len("param."):] for k in z if k.startswith("param."))
    assert sorted(sd.keys()) == ref_keys
    for k in ref_keys:
        if ".norm" in k and (k.endswith("weight") or k.endswith("bias")):
            continue  # re-drawn by the fixture generator after construction
        tol = 2e-6 if (".V" in k) else 0.0
        np.testing.assert_allclose(sd[k].numpy(), z["param." + k], rtol=0, atol=tol, err_msg=k)
    assert net.is_snn is False
    with pytest.raises(ValueError):
        ANN((4, None, 8), [8, 4], ann_type="LSTM")
    with pytest.raises(ValueError):
        ANN((4, None, 8), [8, 4], ann_type="MLP", bidirectional=True)


# ------------------------------------------------------------------ f-3: SHD / SSC loader API
def _fake_h5(n=11, seed=3):
    rng = np.random.default_rng(seed)
    times, units = [], []
    for i in range(n):
        m = int(rng.integers(0, 400)) if i != 4 else 0          # one empty sample
        t = np.sort(rng.uniform(0.0, 1.39, m)).astype(np.float32)
        if m > 3:
            t[:2] = 0.0                                           # exactly on the first edge
        times.append(t)
        units.append(rng.integers(0, 700, m).astype(np.int32))
    return {"spikes": {"times": times, "units": units}, "labels": rng.integers(0, 20, n)}


def test_spiking_dataset_api_and_errors():
    from sparch_amd.dataloaders.spiking_datasets import SpikingDataset, load_shd_or_ssc
    from oracle import events_numpy as ev

    with pytest.raises(ValueError):
        load_shd_or_ssc("mnist", "/x", "train", 4)
    with pytest.raises(ValueError):
        load_shd_or_ssc("shd", "/x", "dev", 4)
    with pytest.raises(ValueError):
        load_shd_or_ssc("shd", "/x", "train", 4, workers=2, h5_file=_fake_h5())
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            SpikingDataset("shd", "/nonexistent", "train")
    ds = SpikingDataset("ssc", "/unused", "valid", nb_steps=100, h5_file=_fake_h5(), device="cpu")
    assert len(ds) == 11 and ds.nb_units == 700 and ds.max_time == 1.4
    t, u, y = ds[2]
    assert t.dtype == np.float32 and u.dtype == np.int32 and isinstance(y, int)
    # dense_sample restates the reference's __getitem__ (spiking_datasets.py:66-78): check against the oracle
    for i in (0, 2, 4):
        x, yy = ds.dense_sample(i)
        np.testing.assert_array_equal(x.numpy(), ev.bin_sample(ds.firing_times[i], ds.units_fired[i])[0])
        assert yy == int(ds.labels[i])
    # the import path of the reference resolves to the same classes
    from sparch.dataloaders.spiking_datasets import SpikingDataset as S2
    assert S2 is SpikingDataset


def test_initial_state_draws_are_torch_rand_bit_for_bit():
    """The host routine behind SNN.draw_states (sparch_mt19937_uniform_f32 on the generator's serialized state)
    must give the numbers torch.rand gives, in the same order, and leave the global generator where torch.rand
    would have left it (the next draw of anything else is unchanged) — from any position inside a block."""
    from sparch_amd import snns

    assert snns._fast_rand_available(), "layout check against torch.rand failed: the fast path is off"
    shapes = [(256, 1024), (256, 1024), (256, 1024), (64, 35), (3, 5), (1, 1)]
    for warm in (0, 1, 623, 624, 625, 1000):
        torch.manual_seed(1234 + warm)
        torch.rand(warm) if warm else None
        want = [torch.rand(r, c) for r, c in shapes]
        tail_want = torch.randn(11)  # a different kind of draw afterwards
        torch.manual_seed(1234 + warm)
        torch.rand(warm) if warm else None
        got = snns._rand_batch(shapes, torch.device("cpu"))
        tail_got = torch.randn(11)
        for a, b in zip(want, got):
            assert torch.equal(a, b)
        assert torch.equal(tail_want, tail_got)


def test_snn_draw_states_order_and_fallback(monkeypatch):
    """SNN.draw_states: one batch in the reference's order (per hidden layer u, [w], s, then the readout's u);
    the torch.rand fallback (SPARCH_FAST_RAND=0 / failed layout check) gives the same tensors."""
    from sparch_amd import snns

    def draw(fast):
        monkeypatch.setattr(snns, "_fast_rand_ok", fast)
        net = snns.SNN((6, None, 20), [32, 32, 5], neuron_type="RadLIF", dropout=0.0, normalization="none",
                       bidirectional=True)
        torch.manual_seed(77)
        st = net.draw_states(6, torch.device("cpu"))
        return st, torch.rand(3)

    (a, ta), (b, tb) = draw(True), draw(False)
    torch.manual_seed(77)
    ref = [torch.rand(12, 32) for _ in range(6)] + [torch.rand(6, 5)]
    flat = lambda st: [t for x in st for t in (x if isinstance(x, tuple) else (x,))]  # noqa: E731
    for x, y, z in zip(flat(a), flat(b), ref):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert torch.equal(ta, tb)


def test_bench_roofline_traffic_comes_from_the_profile_of_the_workload_run():
    """bench.py's `roofline.traffic` is read from the tracked PMC passes (bench.py cannot profile itself): the file
    must be the one of the workload and precision actually run — a cfg4 / cfg5 line carried the headline workload's
    bytes before round 3 fixed the lookup — and carry the commit it was collected on; no file, no number."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    by_wl = {}
    for wl, low in (("cfg3", False), ("cfg3", True), ("cfg4", False), ("cfg5", True)):
        val, src = bench.pmc_traffic("rec_cell_bwd[RadLIF]", wl, low)
        assert val is not None and val > 1e8, (wl, low)
        assert src["file"] == f"profiles/r03_pmc_traffic_{wl}{'_bf16' if low else ''}.json" and src["commit"]
        by_wl[(wl, low)] = val
    assert len(set(by_wl.values())) == len(by_wl)            # four workloads, four different byte counts
    assert by_wl[("cfg3", False)] < 1.6 * 16 * 1024 * 64000  # the backward kernel within 1.6x its algorithmic bytes
    val, src = bench.pmc_traffic("cell_bwd[adLIF]", "cfg2", False)
    assert val is not None and src["file"].endswith("cfg2.json")
    assert bench.pmc_traffic("rec_cell_bwd[RadLIF]", "cfg5", False) == (None, None)  # no fp32 pass of cfg5 is tracked
    assert bench.pmc_traffic("gemm_nn[64000x1024x1024]", "mlp", False) == (None, None)
