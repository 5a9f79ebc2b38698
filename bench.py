#!/usr/bin/env python3
"""
bench.py — BASELINE.json metric: train-step timesteps*samples/sec, RadLIF 3x1024 on synthetic
SSC-shaped input (B=256 per GPU, T=250, C=700), fp32, at 1/2/4/8 MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the training hot path over one resident synthetic batch:
zero_grad -> SNN.forward -> cross-entropy on the softmax-sum -> backward (-> per-layer RCCL
gradient all-reduce, overlapped) -> Adam.step   (reference: exp.py:359-377).
Inputs are in HBM before the timed region.  N>1 is weak scaling: every rank runs the full
B=256 batch (data parallel, no data-path collective; only the gradient all-reduce).

Rank 0 prints ONE JSON line with the contract keys plus
  "roofline":     dominant kernel's achieved algorithmic rate vs the gfx950 peak (HIP events, live)
  "cpu_baseline": the CPU oracle (a validated restatement of the reference's eager PyTorch path,
                  oracle/snn_oracle.py) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import re
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak (6.29 TB/s measured achievable)

WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on (default)
    "cfg3": dict(neuron_type="RadLIF", layer_sizes=[1024, 1024, 35], B=256, T=250, C=700, pdrop=0.1),
    # BASELINE.json configs[1]: adLIF 3x512 on SHD shapes (HBM-bound fused cell kernels); informational
    "cfg2": dict(neuron_type="adLIF", layer_sizes=[512, 512, 20], B=128, T=250, C=700, pdrop=0.1),
    # SURVEY f-4: the reference's non-spiking baselines at the headline shape; informational
    "rnn": dict(neuron_type="RNN", layer_sizes=[1024, 1024, 35], B=256, T=250, C=700, pdrop=0.1),
    "mlp": dict(neuron_type="MLP", layer_sizes=[1024, 1024, 35], B=256, T=250, C=700, pdrop=0.1),
    "ligru": dict(neuron_type="LiGRU", layer_sizes=[1024, 1024, 35], B=256, T=250, C=700, pdrop=0.1),
    "gru": dict(neuron_type="GRU", layer_sizes=[1024, 1024, 35], B=256, T=250, C=700, pdrop=0.1),
}
WORKLOAD = WORKLOADS["cfg3"]


def algorithmic_work(name, B, T, H):
    """(bound, amount per launch, unit) for a timed call — SURVEY.md §8(d) figures, stated in DESIGN.md."""
    m = re.match(r"gemm_(?:spike_|auto_)?(nt|nn|tn)\[(\d+)x(\d+)x(\d+)\]", name)
    if m:
        M, N, K = int(m.group(2)), int(m.group(3)), int(m.group(4))
        return "mfma", 2.0 * M * N * K, "flop"
    if name.startswith("rec_cell_fwd") or name.startswith("rec_cell_bwd") or name.startswith("ann_rec_"):
        # one (B x H) x (H x H) recurrent product per time step (s_{t-1} V  or  dWx_{t+1} V^T)
        return "mfma", 2.0 * B * H * H * T, "flop"
    if name.startswith("cell_fwd") or name.startswith("cell_bwd"):
        per = 16 if "adLIF" in name else 12  # bytes per neuron-step: Wx + s + u (+ w) | g + u (+ w) + dWx
        return "hbm", float(per) * B * T * H, "byte"
    return None


def pmc_traffic(kernel_label):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs of this same workload, gfx950
    FETCH_SIZE correction applied).  bench.py cannot run the profiler on itself; None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    want = {"rec_cell_bwd": "rec_bwd_kernel", "rec_cell_fwd": "rec_fwd_kernel"}
    for prefix, kname in want.items():
        if kernel_label.startswith(prefix):
            for k, v in table.items():
                if k.startswith(kname):
                    return v["hbm_bytes"]
    return None


def cpu_baseline(rank):
    """The oracle on the host cores, bounded sample of the same workload (same model, B=64 of 256)."""
    from oracle import snn_oracle as orc

    try:
        n_threads = len(os.sched_getaffinity(0))
    except AttributeError:
        n_threads = os.cpu_count() or 1
    n_threads = max(1, min(n_threads, 16))  # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(n_threads)
    Bs, T, C = 64, WORKLOAD["T"], WORKLOAD["C"]
    sizes = WORKLOAD["layer_sizes"]
    g = torch.Generator().manual_seed(1234)
    H = sizes[0]
    p = {}
    fan = C
    for i in range(2):
        p[f"snn.{i}.W.weight"] = (torch.rand(H, fan, generator=g) * 2 - 1) / fan ** 0.5
        p[f"snn.{i}.V.weight"] = torch.nn.init.orthogonal_(torch.empty(H, H), generator=g)
        p[f"snn.{i}.alpha"] = torch.rand(H, generator=g) * 0.14 + 0.82
        p[f"snn.{i}.beta"] = torch.rand(H, generator=g) * 0.024 + 0.967
        p[f"snn.{i}.a"] = torch.rand(H, generator=g) * 2 - 1
        p[f"snn.{i}.b"] = torch.rand(H, generator=g) * 2
        p[f"snn.{i}.norm.weight"], p[f"snn.{i}.norm.bias"] = torch.ones(H), torch.zeros(H)
        p[f"snn.{i}.norm.running_mean"], p[f"snn.{i}.norm.running_var"] = torch.zeros(H), torch.ones(H)
        fan = H
    Co = sizes[-1]
    p["snn.2.W.weight"] = (torch.rand(Co, H, generator=g) * 2 - 1) / H ** 0.5
    p["snn.2.alpha"] = torch.rand(Co, generator=g) * 0.14 + 0.82
    p["snn.2.norm.weight"], p["snn.2.norm.bias"] = torch.ones(Co), torch.zeros(Co)
    p["snn.2.norm.running_mean"], p["snn.2.norm.running_var"] = torch.zeros(Co), torch.ones(Co)
    for k, v in p.items():
        if "running" not in k:
            v.requires_grad_(True)
    x = (torch.rand(Bs, T, C, generator=g) < 0.05).float()
    y = torch.randint(0, Co, (Bs,), generator=g)
    torch.manual_seed(7)
    init = orc.draw_init_states(Bs, sizes, "RadLIF")
    t0 = time.perf_counter()
    out, rates = orc.snn_forward(x, p, neuron_type="RadLIF", num_layers=3, init_states=init, training=True,
                                 stats={})
    loss = orc.train_step_loss(out, rates, y)
    loss.backward()
    dt = time.perf_counter() - t0
    return {"value": Bs * T / dt, "unit": "timesteps*samples/s", "cores": n_threads, "kind": "port",
            "sample": f"1 fwd+bwd of the same RadLIF [1024,1024,35] model at B={Bs} (of 256), T={T}, C={C}, "
                      f"pdrop=0, eager torch CPU oracle, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg3")
    args = ap.parse_args()
    global WORKLOAD
    WORKLOAD = WORKLOADS[args.workload]

    import sparch_amd
    from sparch_amd import dp
    from sparch_amd import functional as Fn

    # RCCL ("nccl") is the real backend; SPARCH_DIST_BACKEND=gloo + SPARCH_SHARE_GPU=1 exist only to rehearse the
    # multi-process code path on a one-GPU box (all ranks on cuda:0)
    backend = os.environ.get("SPARCH_DIST_BACKEND", "nccl")
    share = os.environ.get("SPARCH_SHARE_GPU", "0") == "1"
    if share:
        torch.cuda.set_device(0)
    rank, world, local = dp.init_from_env(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    w = WORKLOAD
    B, T, C, H = w["B"], w["T"], w["C"], w["layer_sizes"][0]
    torch.manual_seed(1234)
    if w["neuron_type"] in ("RNN", "MLP", "LiGRU", "GRU"):
        from sparch_amd.anns import ANN
        net = ANN((B, None, C), w["layer_sizes"], ann_type=w["neuron_type"], dropout=w["pdrop"],
                  normalization="batchnorm").to(dev)
    else:
        net = sparch_amd.SNN((B, None, C), w["layer_sizes"], neuron_type=w["neuron_type"], dropout=w["pdrop"],
                             normalization="batchnorm").to(dev)
    net.train()
    opt = sparch_amd.optim.Adam(net.parameters(), 1e-2)  # exp.py:89 (same arithmetic, one launch: SURVEY f-2)
    loss_fn = torch.nn.CrossEntropyLoss()           # exp.py:100
    reducer = dp.GradAllReducer(net) if world > 1 else None
    g = torch.Generator().manual_seed(4321 + rank)
    x = (torch.rand(B, T, C, generator=g) < 0.05).float().to(dev)
    y = torch.randint(0, w["layer_sizes"][-1], (B,), generator=g).to(dev)
    torch.manual_seed(99 + rank)

    def step():
        opt.zero_grad(set_to_none=True)
        out, rates = net(x)
        loss = loss_fn(out, y)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    Fn.check_status(dev)
    Fn.timer.reset()
    Fn.timer.enabled = True
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    Fn.timer.enabled = False
    totals = Fn.timer.collect()
    Fn.check_status(dev)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    final_loss = float(loss.item())

    if rank == 0:
        ms = 1e3 * dt / args.steps
        value = world * B * T * args.steps / dt
        # dominant kernel by total time inside the timed region
        kern = {k: {"launches": c, "avg_ms": t / c} for k, (c, t) in totals.items()}
        dom = max(totals.items(), key=lambda kv: kv[1][1])[0] if totals else None
        roof = None
        work = algorithmic_work(dom, B, T, H) if dom is not None else None
        if work is not None:
            bound, amount, _ = work
            avg_s = kern[dom]["avg_ms"] * 1e-3
            if bound == "mfma":
                ach = amount / avg_s / 1e12
                roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_MFMA_F32_TFLOPS,
                        "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_F32_TFLOPS, "traffic": pmc_traffic(dom),
                        "avg_ms": kern[dom]["avg_ms"]}
            else:
                ach = amount / avg_s / 1e9
                roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": ach / PEAK_HBM_GBS, "traffic": None, "avg_ms": kern[dom]["avg_ms"]}
        print(f"[bench] gpu: {value:.0f} ts*samples/s, {ms:.2f} ms/step; dominant {dom}; "
              f"timing the CPU oracle sample next", file=sys.stderr, flush=True)
        cpu = None
        if world == 1 and not args.no_cpu_baseline and args.workload == "cfg3":
            cpu = cpu_baseline(rank)
        line = {
            "metric": "train-step timesteps*samples/sec (fwd+bwd+Adam), " +
                      {"cfg3": "RadLIF 3x1024 SSC shape", "cfg2": "adLIF 3x512 SHD shape",
                       "rnn": "RNN baseline 3x1024 SSC shape", "mlp": "MLP baseline 3x1024 SSC shape",
                       "ligru": "LiGRU baseline 3x1024 SSC shape", "gru": "GRU baseline 3x1024 SSC shape"}[args.workload],
            "value": value, "unit": "timesteps*samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{w['neuron_type']} {w['layer_sizes']} batchnorm pdrop={w['pdrop']}, B={B}/GPU "
                                   f"T={T} C={C} Bernoulli(0.05) spikes (BASELINE.json "
                                   f"configs[{2 if args.workload == 'cfg3' else 1}])",
                       "global_batch": world * B, "seq_len": T, "parallelism": f"dp{world}"},
            "roofline": roof, "cpu_baseline": cpu,
            "kernels_ms_per_step": {k: round(v["avg_ms"] * v["launches"] / args.steps, 4) for k, v in kern.items()},
            "final_loss": final_loss,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
