#!/usr/bin/env python3
"""One rank of the two-rank data-parallel check on ONE GPU (tests/test_a_multirank_gpu.py starts two of these
before anything in the test session touches the GPU).  Rendezvous over gloo on 127.0.0.1, every rank on cuda:0
(SPARCH_SHARE_GPU=1).  Each rank
  1. runs a real RadLIF training step on its half of a global batch through sparch_amd.dp.GradAllReducer (both
     launch policies), and checks the averaged gradients against the mean of the two shards' gradients that it
     computes alone (same kernels, same seeded initial states -> equal up to the order of the final sum);
  2. with SyncBN (functional.SYNC_BN) runs an adLIF net on its half and checks that its output rows, the
     running statistics and the reduced gradients match ONE process running the whole batch.
Exit code 0 = all checks passed on this rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparch_amd  # noqa: E402
from sparch_amd import dp  # noqa: E402
from sparch_amd import functional as Fn  # noqa: E402


def relmax(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-6))


def grads_of(net, x, y, seed, reducer=None, check=True):
    net.zero_grad(set_to_none=True)
    for lay in net.snn:
        lay._calls = 0
    torch.manual_seed(seed)  # the layers draw their initial states from the CPU generator
    out, rates = net(x)
    torch.nn.functional.cross_entropy(out, y).backward()
    if reducer is not None:
        reducer.finish()
    if check:
        Fn.check_status()
    return out.detach().clone(), {k: v.grad.detach().clone() for k, v in net.named_parameters()}


def main():
    rank, world, _ = dp.init_from_env()
    assert world == 2 and dist.get_backend() == "gloo"
    dev = torch.device("cuda", 0)
    B, T, C = 16, 30, 40
    g = torch.Generator().manual_seed(11)
    x_all = (torch.rand(B, T, C, generator=g) < 0.2).float().to(dev)
    y_all = torch.randint(0, 20, (B,), generator=g).to(dev)
    xs, ys = dp.shard_batch(x_all, rank, world), dp.shard_batch(y_all, rank, world)

    # ---- 1. gradient all-reduce on a recurrent model, both policies
    torch.manual_seed(1)
    net = sparch_amd.SNN((B // world, None, C), [64, 64, 20], neuron_type="RadLIF", dropout=0.0).to(dev).train()
    for p in net.parameters():
        dist.broadcast(p.data, src=0)
    assert net.snn[0].uses_persistent_kernel
    shard = []
    for r in range(world):  # what each shard contributes, computed alone (running stats restored afterwards)
        state = {k: v.clone() for k, v in net.state_dict().items()}
        shard.append(grads_of(net, dp.shard_batch(x_all, r, world), dp.shard_batch(y_all, r, world), 100 + r)[1])
        net.load_state_dict(state)
    for policy in dp.POLICIES:
        state = {k: v.clone() for k, v in net.state_dict().items()}
        red = dp.GradAllReducer(net, policy=policy, rows_per_rank=B // world, trace=True)
        _, got = grads_of(net, xs, ys, 100 + rank, red)
        if policy == "window":  # no collective inside or across a persistent launch (tests/test_cli_and_dp.py's rule)
            inflight, inside = set(), False
            for what, i in red.trace:
                if what == "pre":
                    inside = True
                elif what == "post":
                    assert not inflight, red.trace
                    inside = False
                elif what == "launch":
                    assert not inside, red.trace
                    inflight.add(i)
                elif what == "wait":
                    inflight.discard(i)
            assert sum(1 for what, _ in red.trace if what == "post") >= 4  # two recurrent layers, forward and backward
        red.remove()
        net.load_state_dict(state)
        for k in got:
            ref = (shard[0][k] + shard[1][k]) / 2
            e = relmax(got[k], ref)
            assert e <= 1e-6, (policy, k, e)
    # default policy: models on the persistent kernels keep their collectives in the windows between those launches
    r_ = dp.GradAllReducer(net, rows_per_rank=B // world)
    assert r_.policy == "window"
    r_.remove()

    # ---- 1b. the case the policy exists to avoid, FORCED: two ranks whose persistent grids each want every CU of
    #          the one GPU they share (8 row tiles x 32 column tiles = 256 workgroups), collectives launched freely
    #          under backward (policy "overlap" = SPARCH_DP_OVERLAP=1).  Whatever the hardware does with two such
    #          grids — run them one after the other, or starve one until its bounded spin gives up — the step
    #          must END: a timeout on either rank reaches both through the collective status word, both switch
    #          to one launch per time step, repeat the step, and the averaged gradients are right.
    Bb, Tb = 256, 12
    gb = torch.Generator().manual_seed(21)
    xb = (torch.rand(world * Bb, Tb, C, generator=gb) < 0.2).float().to(dev)
    yb = torch.randint(0, 20, (world * Bb,), generator=gb).to(dev)
    torch.manual_seed(3)
    big = sparch_amd.SNN((Bb, None, C), [1024, 1024, 20], neuron_type="RadLIF", dropout=0.0).to(dev).train()
    for p in big.parameters():
        dist.broadcast(p.data, src=0)
    state = {k: v.clone() for k, v in big.state_dict().items()}
    red = dp.GradAllReducer(big, policy="overlap", rows_per_rank=Bb)
    got, tries = None, 0
    for tries in range(1, 4):
        big.load_state_dict(state)
        _, got = grads_of(big, dp.shard_batch(xb, rank, world), dp.shard_batch(yb, rank, world), 300 + rank, red,
                          check=False)
        if not Fn.poll_status(dev):  # (finish() made the word collective: both ranks take the same branch)
            break
        print(f"rank {rank}: forced overlap on a full grid, attempt {tries}: {Fn.describe_timeout()} -> "
              f"per-step launches on every rank", flush=True)
        Fn.degrade(dev)
    else:
        raise AssertionError("the step still times out after degrading to per-step launches")
    red.remove()
    # the shards' own gradients, for comparison: with one launch per time step (nothing waits inside a launch) —
    # two processes' full-grid persistent kernels on ONE GPU are exactly what can time out, with or without a
    # collective (seen in round 3: "rec_fwd gave up at time step 2" in this very loop; profiles/r03_dp_full_grid_*)
    Fn.degrade(dev)
    shard_b = []
    for r in range(world):
        big.load_state_dict(state)
        shard_b.append(grads_of(big, dp.shard_batch(xb, r, world), dp.shard_batch(yb, r, world), 300 + r)[1])
    for k in got:
        e = relmax(got[k], (shard_b[0][k] + shard_b[1][k]) / 2)
        assert e <= 2e-6, ("forced overlap", k, e)
    print(f"rank {rank}: forced overlap on a full grid finished after {tries} attempt(s), gradients correct "
          f"(timed out and degraded on the way: {tries > 1})", flush=True)
    Fn._degraded.clear()
    del big, red, shard_b, got

    # ---- 1c. a timeout on ONE rank: rank 1's status word is raised by hand between backward and the reducer.  The
    #          reducer makes the word collective, so BOTH ranks' optimizer steps are no-ops on the device (the peers
    #          had averaged rank 1's invalid gradients in), both count the skipped step, both take it back from
    #          Adam's bias-correction counter when they read the word — and the replicas stay identical.
    torch.manual_seed(4)
    net = sparch_amd.SNN((B // world, None, C), [64, 64, 20], neuron_type="RadLIF", dropout=0.0).to(dev).train()
    for p in net.parameters():
        dist.broadcast(p.data, src=0)
    opt = sparch_amd.optim.Adam(net.parameters(), 1e-2)
    red = dp.GradAllReducer(net, rows_per_rank=B // world)

    def train_step(seed, poison):
        opt.zero_grad(set_to_none=True)
        torch.manual_seed(seed)
        out, _ = net(xs)
        torch.nn.functional.cross_entropy(out, ys).backward()
        if poison:
            Fn.status_word(dev)[0] = 1
        red.finish()
        opt.step()

    def replicas_equal():
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        return bool(torch.equal(both[0], both[1])), flat

    train_step(500 + rank, False)
    ok, before = replicas_equal()
    assert ok and not Fn.poll_status(dev)
    train_step(501 + rank, poison=(rank == 1))
    ok, after = replicas_equal()
    assert ok, "replicas diverged after a timeout on one rank"
    assert torch.equal(before, after), "a step with a raised status word must not move the parameters on ANY rank"
    assert Fn.poll_status(dev), f"rank {rank} did not see the peer's timeout"
    assert Fn.last_timeout["skipped_steps"] == 1
    steps = {float(opt.state[p]["step"]) for p in net.parameters()}
    assert steps == {1.0}, steps  # the skipped step was taken back from the bias-correction counter
    train_step(502 + rank, False)
    ok, later = replicas_equal()
    assert ok and not torch.equal(later, after) and not Fn.poll_status(dev)
    red.remove()
    print(f"rank {rank}: a timeout on one rank skips the step on both; replicas identical", flush=True)

    # ---- 2. SyncBN: two ranks x B/2 rows == one process x B rows
    torch.manual_seed(2)
    net = sparch_amd.SNN((B, None, C), [48, 48, 20], neuron_type="adLIF", dropout=0.0).to(dev).train()
    for p in net.parameters():
        dist.broadcast(p.data, src=0)
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def states_for(rows):  # identical initial states for the same global rows in both runs
        gen = torch.Generator().manual_seed(5)
        full = [torch.rand(B, 48, generator=gen) for _ in range(6)] + [torch.rand(B, 20, generator=gen)]
        return iter([t[rows].contiguous() for t in full])

    from sparch_amd import snns as snn_mod
    real_rand = snn_mod._rand_to

    def run(x, y, rows, sync):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        it = states_for(rows)
        snn_mod._rand_to = lambda r, c, device: next(it).to(device)
        Fn.SYNC_BN = {"group": None, "world": world} if sync else None
        try:
            out, rates = net(x)
            (torch.nn.functional.cross_entropy(out, y, reduction="sum") / B).backward()
        finally:
            snn_mod._rand_to = real_rand
            Fn.SYNC_BN = None
        Fn.check_status()
        return (out.detach().clone(), {k: v.grad.detach().clone() for k, v in net.named_parameters()},
                {k: v.clone() for k, v in net.state_dict().items() if "running" in k})

    rows = slice(rank * B // world, (rank + 1) * B // world)
    out_full, g_full, st_full = run(x_all, y_all, slice(0, B), False)
    out_dp, g_dp, st_dp = run(xs, ys, rows, True)
    flat = torch.cat([v.reshape(-1) for v in g_dp.values()])
    dist.all_reduce(flat)  # loss is normalised by the GLOBAL batch above: the shards' gradients add up
    off = 0
    # the batch statistics agree up to summation order, so a membrane potential within rounding of the
    # threshold may flip: allow a handful of differing output entries, none expected
    assert float((out_dp - out_full[rows]).abs().max()) <= 2e-3 * T
    for k, v in st_dp.items():
        np.testing.assert_allclose(v.cpu().numpy(), st_full[k].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)
    for k, v in g_dp.items():
        n = v.numel()
        e = relmax(flat[off:off + n].view_as(v), g_full[k])
        off += n
        assert e <= 5e-3, (k, e)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: data-parallel checks passed", flush=True)


if __name__ == "__main__":
    main()
