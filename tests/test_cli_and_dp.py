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
        assert red.overlap is True  # no persistent-kernel layers -> per-layer overlapped launches
        net.snn[1].V = torch.nn.Identity()  # a recurrent matrix alone (LiGRU / GRU: launch per step) changes nothing
        assert dp.GradAllReducer(net).overlap is True
        # a layer on the persistent recurrent kernel: deferred when its grid fills the GPU (or the batch is
        # unknown), overlapped when the per-rank batch leaves CUs to RCCL
        net.snn[1].uses_persistent_kernel, net.snn[1].hidden_size = True, 1024
        assert dp.GradAllReducer(net).overlap is False
        assert dp.GradAllReducer(net, rows_per_rank=256).overlap is False
        assert dp.GradAllReducer(net, rows_per_rank=128).overlap is True
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


def test_shard_batch_errors():
    with pytest.raises(ValueError):
        dp.shard_batch(torch.zeros(7, 3), 0, 2)
