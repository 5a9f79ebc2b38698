"""CPU: CLI flag surface (reference model_config.py:19-65, training_config.py:19-147) and the
data-parallel gradient all-reducer on a 2-process gloo group."""
import argparse
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sparch_amd import dp, parsers

# name -> default, as the reference's argparse declares them
REF_MODEL = {"model_type": "LIF", "nb_layers": 3, "nb_hiddens": 128, "pdrop": 0.1, "normalization": "batchnorm",
             "use_bias": False, "bidirectional": False}
REF_TRAIN = {"use_pretrained_model": False, "only_do_testing": False, "load_exp_folder": None,
             "new_exp_folder": None, "dataset_name": "shd", "data_folder": "data/shd_dataset/", "log_tofile": False,
             "save_best": True, "batch_size": 128, "nb_epochs": 5, "start_epoch": 0, "lr": 1e-2,
             "scheduler_patience": 1, "scheduler_factor": 0.7, "use_regularizers": False, "reg_factor": 0.5,
             "reg_fmin": 0.01, "reg_fmax": 0.5, "use_augm": False}


def _parser():
    p = argparse.ArgumentParser()
    parsers.add_model_options(p)
    parsers.add_training_options(p)
    return p


def test_flag_names_and_defaults_match_reference():
    ns = vars(_parser().parse_args([]))
    for k, v in {**REF_MODEL, **REF_TRAIN}.items():
        assert ns[k] == v, k
    assert set(ns) - set(REF_MODEL) - set(REF_TRAIN) == {"synthetic", "synthetic_batches", "seq_len", "sync_bn",
                                                             "compute_dtype"}


def test_flag_parsing_booleans_and_choices():
    ns = _parser().parse_args(["--model_type", "RadLIF", "--bidirectional", "True", "--use_bias", "yes",
                               "--save_best", "0", "--dataset_name", "ssc", "--pdrop", "0.25"])
    assert ns.model_type == "RadLIF" and ns.bidirectional is True and ns.use_bias is True
    assert ns.save_best is False and ns.dataset_name == "ssc" and ns.pdrop == 0.25
    with pytest.raises(SystemExit):
        _parser().parse_args(["--model_type", "LSTM"])
    with pytest.raises(SystemExit):
        _parser().parse_args(["--dataset_name", "mnist"])
    with pytest.raises(ValueError):
        parsers.strtobool("maybe")


def test_run_exp_module_and_shims_import():
    import run_exp
    from sparch.exp import Experiment  # noqa: F401
    from sparch.parsers.model_config import add_model_options  # noqa: F401
    from sparch.parsers.training_config import add_training_options  # noqa: F401
    ns = run_exp.parse_args(["--nb_hiddens", "64"])
    assert ns.nb_hiddens == 64


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.snn = torch.nn.ModuleList([torch.nn.Linear(6, 5), torch.nn.Linear(5, 4), torch.nn.Linear(4, 3)])

    def forward(self, x):
        for lay in self.snn:
            x = torch.tanh(lay(x))
        return x


def _dp_worker(rank, world, port, q, overlap=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = dp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    net = _Toy()
    if overlap is None:
        red = dp.GradAllReducer(net)
        assert red.policy == "overlap"  # no persistent-kernel layers -> per-layer overlapped launches
        net.snn[1].V = torch.nn.Identity()  # a recurrent matrix alone (LiGRU / GRU: launch per step) changes nothing
        assert dp.GradAllReducer(net).policy == "overlap"
        # a layer on the persistent recurrent kernel: collectives only in the windows between persistent launches,
        # at any per-rank batch (rounds 1-2 overlapped freely when the grid left 32 CUs: a guess about RCCL)
        net.snn[1].uses_persistent_kernel, net.snn[1].hidden_size = True, 1024
        for rows in (None, 256, 128):
            r_ = dp.GradAllReducer(net, rows_per_rank=rows)
            assert r_.policy == "window" and r_.overlap is True
            r_.remove()
        os.environ["SPARCH_DP_OVERLAP"] = "0"
        assert dp.GradAllReducer(net).policy == "deferred"
        os.environ["SPARCH_DP_OVERLAP"] = "1"
        assert dp.GradAllReducer(net).policy == "overlap"
        del os.environ["SPARCH_DP_OVERLAP"]
        os.environ["SPARCH_DP_POLICY"] = "deferred"
        assert dp.GradAllReducer(net).policy == "deferred"
        del os.environ["SPARCH_DP_POLICY"]
        del net.snn[1].V, net.snn[1].uses_persistent_kernel, net.snn[1].hidden_size
    else:
        red = dp.GradAllReducer(net, overlap=overlap)
    assert len(red.buckets) == 3 and red.bytes_per_step == sum(p.numel() for p in net.parameters()) * 4
    g = torch.Generator().manual_seed(5)
    x_all = torch.randn(8, 6, generator=g)
    out = []
    for step in range(2):  # two steps: buckets must re-arm
        net.zero_grad()
        x = dp.shard_batch(x_all + step, rank, world)
        net(x).pow(2).sum().backward()
        red.finish()
        out.append([p.grad.clone() for p in net.parameters()])
    dist.barrier()
    if rank == 0:
        q.put([[t.numpy() for t in o] for o in out])
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [None, False])
def test_grad_allreducer_gloo_world2(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket

    with socket.socket() as sock:  # a port nobody is listening on right now
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)  # a rank that died must fail the test, not hang it
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference: mean over the two shards' gradients
    torch.manual_seed(0)
    net = _Toy()
    g = torch.Generator().manual_seed(5)
    x_all = torch.randn(8, 6, generator=g)
    for step in range(2):
        ref = None
        for r in range(2):
            net.zero_grad()
            net(dp.shard_batch(x_all + step, r, 2)).pow(2).sum().backward()
            gr = [p.grad.clone() for p in net.parameters()]
            ref = gr if ref is None else [a + b for a, b in zip(ref, gr)]
        for a, b in zip(got[step], ref):
            assert torch.allclose(torch.from_numpy(a), b / 2, rtol=1e-6, atol=1e-7)


class _PersistentSim(torch.autograd.Function):
    """Identity whose forward and backward each enqueue one 'persistent launch' the way sparch_amd.functional does."""

    @staticmethod
    def forward(ctx, x):
        from sparch_amd import functional as Fn
        with Fn._persistent_launch():
            pass
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        from sparch_amd import functional as Fn
        with Fn._persistent_launch():
            pass
        return g


class _ToyRecurrent(_Toy):
    """Two hidden layers on 'persistent kernels' (one launch per pass, then the layer's ordinary work), a readout."""

    def __init__(self):
        super().__init__()
        for lay in self.snn[:2]:
            lay.uses_persistent_kernel, lay.hidden_size = True, 1024

    def forward(self, x):
        for i, lay in enumerate(self.snn):
            x = lay(x)
            x = torch.tanh(_PersistentSim.apply(x) if i < 2 else x)
        return x


def _window_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dp.init_from_env("gloo")
    torch.manual_seed(0)
    net = _ToyRecurrent()
    red = dp.GradAllReducer(net, trace=True)
    assert red.policy == "window"
    g = torch.Generator().manual_seed(5)
    x_all = torch.randn(8, 6, generator=g)
    out, traces = [], []
    for step in range(2):
        net.zero_grad()
        del red.trace[:]
        net(dp.shard_batch(x_all + step, rank, world)).pow(2).sum().backward()
        red.finish()
        out.append([p.grad.clone() for p in net.parameters()])
        traces.append(list(red.trace))
    dist.barrier()
    if rank == 0:
        q.put(([[t.numpy() for t in o] for o in out], traces))
    dist.destroy_process_group()


def test_window_policy_keeps_collectives_out_of_persistent_launches():
    """sparch_amd.dp "window": with layers on the persistent recurrent kernels every bucket's all-reduce is enqueued
    right behind the NEXT persistent launch and waited for right before the one after it.  On a 2-process gloo
    group: the gradients are the mean of the shards', and in the recorded event order (a) nothing is launched
    between a launch's `pre` and `post`, (b) nothing is in flight when a persistent launch is enqueued, (c) the
    readout's and the second layer's buckets go out during backward (behind layer 1's / layer 0's launch), the
    input layer's in finish()."""
    import socket

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = [ctx.Process(target=_window_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, traces = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = _ToyRecurrent()
    g = torch.Generator().manual_seed(5)
    x_all = torch.randn(8, 6, generator=g)
    for step in range(2):
        ref = None
        for r in range(2):
            net.zero_grad()
            net(dp.shard_batch(x_all + step, r, 2)).pow(2).sum().backward()
            gr = [p.grad.clone() for p in net.parameters()]
            ref = gr if ref is None else [a + b for a, b in zip(ref, gr)]
        for a, b in zip(got[step], ref):
            assert torch.allclose(torch.from_numpy(a), b / 2, rtol=1e-6, atol=1e-7)
    for tr in traces:
        inflight, inside = set(), False
        for what, i in tr:
            if what == "pre":
                inside = True
            elif what == "post":
                assert not inflight, f"collectives {inflight} in flight across a persistent launch: {tr}"
                inside = False
            elif what == "launch":
                assert not inside, f"a collective was enqueued inside a persistent launch: {tr}"
                inflight.add(i)
            elif what == "wait":
                inflight.discard(i)
        assert not inflight
        launches = [i for what, i in tr if what == "launch"]
        assert launches == [2, 1, 0], tr  # readout, layer 1, layer 0
        # the readout's bucket leaves behind layer 1's backward launch and is waited for before layer 0's
        k = tr.index(("launch", 2))
        assert tr[k - 1][0] == "post" and ("wait", 2) in tr[k:tr.index(("launch", 1))]
        # the input layer's bucket has no persistent launch behind it: it goes out in finish(), after the last post
        assert tr.index(("launch", 0)) > max(j for j, e in enumerate(tr) if e[0] == "post")


def test_shard_batch_errors():
    with pytest.raises(ValueError):
        dp.shard_batch(torch.zeros(7, 3), 0, 2)
