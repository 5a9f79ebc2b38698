#!/usr/bin/env python3
"""
bench.py — BASELINE.json metric: train-step timesteps*samples/sec, RadLIF 3x1024 on synthetic
SSC-shaped input (B=256 per GPU, T=250, C=700), fp32, at 1/2/4/8 MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the training hot path over one resident synthetic batch:
zero_grad -> SNN.forward -> cross-entropy on the softmax-sum -> backward (-> per-layer RCCL
gradient all-reduce, overlapped) -> Adam.step   (reference: exp.py:359-377).
Inputs are in HBM before the timed region.  N>1 defaults to weak scaling: every rank runs the full
B=256 batch (data parallel, no data-path collective; only the gradient all-reduce); `--scaling strong`
keeps the GLOBAL batch at B and gives every rank B/N rows (SURVEY.md §8e asks for both curves).

Other workloads (`--workload`): cfg2 / cfg4 / cfg5 are BASELINE.json configs[1] / [3] / [4] (cfg4 feeds raw
audio through the HIP mel filterbank inside the step), rnn / mlp / ligru / gru the reference's non-spiking
baselines at the headline shape (SURVEY f-4).  The driver's default line is cfg3.

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
    # BASELINE.json configs[3]: RadLIF 3x1024 on SC raw audio -> HIP mel filterbank front-end (1 s clips at
    # 16 kHz -> 98 frames x 40 log-mel bins) -> the network; the front-end runs inside the timed step
    "cfg4": dict(neuron_type="RadLIF", layer_sizes=[1024, 1024, 35], B=256, T=98, C=40, pdrop=0.1, audio=16000),
    # BASELINE.json configs[4]: bidirectional RadLIF 4x1024, T=1000 long-sequence stress (fp32 here, >= bf16)
    "cfg5": dict(neuron_type="RadLIF", layer_sizes=[1024, 1024, 1024, 35], B=256, T=1000, C=700, pdrop=0.1,
                 bidirectional=True),
}
CONFIG_INDEX = {"cfg2": 1, "cfg3": 2, "cfg4": 3, "cfg5": 4}  # position in BASELINE.json "configs"
PEAK_MFMA_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the pipe the split products run on)
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
    if name.startswith("fbank"):
        return "hbm", None, "byte"
    if name.startswith("cell_fwd") or name.startswith("cell_bwd"):
        per = 16 if "adLIF" in name else 12  # bytes per neuron-step: Wx + s + u (+ w) | g + u (+ w) + dWx
        return "hbm", float(per) * B * T * H, "byte"
    return None


def bf16_issue_factor(name):
    """bf16 MFMA products issued per algorithmic fp32 product by the exact-split kernels: spike x dense = 3
    planes, dense x dense = 6 cross terms."""
    if name.startswith("rec_cell_fwd") or name.startswith("gemm_spike"):
        return 3.0
    if name.startswith("rec_cell_bwd") or name.startswith("ann_rec_") or name.startswith("gemm_nn") \
            or name.startswith("gemm_tn") or name.startswith("gemm_nt"):
        return 6.0
    return None  # gemm_auto_*: decided on the device (3 for bf16-exact inputs, else 6)


def pmc_traffic(kernel_label, workload="cfg3", low=False):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS workload
    (profiles/r03_pmc_traffic_<workload>[_bf16].json: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 FETCH_SIZE
    correction applied; tools/collect_round3.sh + tools/summarize_round3.sh).  bench.py cannot run the profiler on
    itself; None where no pass of the workload is tracked."""
    name = f"r03_pmc_traffic_{workload}{'_bf16' if low else ''}.json"
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        table = json.load(f)
    src = {"file": "profiles/" + name, "commit": table.get("_meta", {}).get("commit")}
    want = {"rec_cell_bwd": "rec_bwd_kernel", "rec_cell_fwd": "rec_fwd_kernel", "cell_bwd": "cell_bwd_pipe_kernel",
            "cell_fwd": "cell_fwd_pipe_kernel"}
    for prefix, kname in want.items():
        if kernel_label.startswith(prefix):
            for k, v in table.items():
                if k.startswith(kname):
                    return v["hbm_bytes"], src
    return None, None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    """(threads to use, size of the affinity mask, cgroup CPU quota or None).  The affinity mask of a GPU box can list
    every CPU of the host (256) while the container's cgroup grants 16: an eager-PyTorch pass on 256 threads inside a
    16-CPU quota takes minutes instead of seconds (seen at the end of round 3).  So: the smaller of the two."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota us> <period us>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = max(1, -(-int(q) // int(per)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                quota = max(1, -(-q // per))
        except (OSError, ValueError):
            pass
    n = max(1, min(affinity, quota) if quota else affinity)
    return n, affinity, quota


def cpu_baseline(state_dict, x_cpu, y_cpu):
    """the oracle (eager PyTorch restatement of the reference's path, pinned to the reference by
    tests/golden) on this host's cores, in the same run: the GPU model's own initial parameters, the first
    B=64 rows of the GPU batch, one warm-up and the median of three timed forward+backward passes in the
    default denormal mode (the reference never sets flush-to-zero), plus one pass with flush-to-zero as a
    footnote.  pdrop = 0 (dropout masks cannot match across devices; the mask multiply is negligible here)."""
    import statistics

    from oracle import snn_oracle as orc

    n_threads, affinity, quota = usable_cpus()
    torch.set_num_threads(n_threads)
    Bs, T, C = 64, WORKLOAD["T"], WORKLOAD["C"]
    sizes = WORKLOAD["layer_sizes"]
    x, y = x_cpu[:Bs].contiguous(), y_cpu[:Bs].contiguous()
    p = {k: v.detach().cpu().clone() for k, v in state_dict.items()}
    for k, v in p.items():
        if v.dtype == torch.float32 and "running" not in k:
            v.requires_grad_(True)

    def one():
        for v in p.values():
            v.grad = None
        torch.manual_seed(7)
        init = orc.draw_init_states(Bs, sizes, "RadLIF")
        t0 = time.perf_counter()
        out, rates = orc.snn_forward(x, p, neuron_type="RadLIF", num_layers=len(sizes), init_states=init,
                                     training=True, stats={})
        orc.train_step_loss(out, rates, y).backward()
        return time.perf_counter() - t0

    warm = one()  # warm-up
    print(f"[bench] cpu oracle: warm-up pass {warm:.1f} s on {n_threads} threads (affinity {affinity}, cgroup quota "
          f"{quota})", file=sys.stderr, flush=True)
    # the leg is bounded: three timed passes when a pass takes what it should (~6 s), one when the host is slow
    times = []
    for _ in range(3 if warm < 12.0 else 1):
        times.append(one())
        print(f"[bench] cpu oracle: pass {len(times)}: {times[-1]:.1f} s", file=sys.stderr, flush=True)
    med = statistics.median(times)
    ftz = float("nan")
    if warm < 12.0:
        torch.set_flush_denormal(True)
        ftz = one()
        torch.set_flush_denormal(False)
    return {"value": Bs * T / med, "unit": "timesteps*samples/s", "cores": n_threads, "kind": "port",
            "rows": Bs, "rows_of_workload": WORKLOAD["B"], "affinity_cpus": affinity, "cgroup_cpu_quota": quota,
            "cpu_model": _cpu_model(), "value_flush_denormals": (Bs * T / ftz) if ftz == ftz else None,
            "sample": f"fwd+bwd of the same RadLIF {sizes} model (the GPU run's initial parameters) on rows 0..{Bs - 1} "
                      f"of the GPU batch (B={Bs} of {WORKLOAD['B']}), T={T}, C={C}, pdrop=0, eager torch CPU oracle: "
                      f"1 warm-up, median of 3 = {med:.1f} s ({', '.join(f'{t:.1f}' for t in times)}); "
                      f"with flush-to-zero {ftz:.1f} s (footnote; the reference runs with denormals on)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 10; cfg2 / cfg4: 200 / 50)")
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed warm-up steps (default: 3; cfg2 / cfg4: 50 / 20 — a ~1 ms step needs tens of ms "
                         "of work before the clocks have settled: cfg2 reads 1.35 ms per step with 5 + 20 steps and "
                         "1.0 ms with 50 + 200, the headline workload 6.5 ms either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg3")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N>1: weak = the workload's batch per GPU, strong = that batch split over the GPUs")
    ap.add_argument("--sync-bn", action="store_true", help="N>1: BatchNorm over the global batch")
    ap.add_argument("--compute-dtype", choices=["fp32", "bf16"], default="fp32",
                    help="operand precision of the matrix products: fp32 = exact bf16 splits (the headline, default); "
                         "bf16 = operands rounded once, fp32 accumulation and state (BASELINE.json configs[4] names bf16)")
    ap.add_argument("--graph", choices=["on", "off", "auto"], default="auto",
                    help="replay the whole step as one captured HIP graph (sparch_amd.graph.GraphedTrainStep); "
                         "auto = when the warm-up shows the host as the bound (enqueue time >= 0.8 x step time)")
    args = ap.parse_args()
    short_step = {"cfg2": (200, 50), "cfg4": (50, 20)}.get(args.workload, (10, 3))
    if args.steps is None:
        args.steps = short_step[0]
    if args.warmup is None:
        args.warmup = short_step[1]
    global WORKLOAD
    WORKLOAD = WORKLOADS[args.workload]

    import sparch_amd
    from sparch_amd import dp
    from sparch_amd import functional as Fn
    Fn.set_compute_dtype(args.compute_dtype)
    low = Fn.compute_dtype() == "bf16"

    # RCCL ("nccl") is the real backend; SPARCH_DIST_BACKEND=gloo + SPARCH_SHARE_GPU=1 exist only to rehearse the
    # multi-process code path on a one-GPU box (all ranks on cuda:0)
    backend = os.environ.get("SPARCH_DIST_BACKEND", "nccl")
    share = os.environ.get("SPARCH_SHARE_GPU", "0") == "1"
    if share:
        torch.cuda.set_device(0)
    rank, world, local = dp.init_from_env(backend)
    # a process group exists: N > 1, or a one-rank group forced on to run this file's distributed branch on RCCL
    # with one GPU (SPARCH_DP_FORCE_COLLECTIVES=1 under torch.distributed.run --nproc-per-node 1)
    multi = dist.is_initialized()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    w = WORKLOAD
    B, T, C, H = w["B"], w["T"], w["C"], w["layer_sizes"][0]
    bidir = bool(w.get("bidirectional", False))
    if args.scaling == "strong":
        if B % world != 0:
            raise SystemExit(f"--scaling strong: global batch {B} is not divisible by {world} GPUs")
        B = B // world  # rows of the global batch on this rank
    torch.manual_seed(1234)
    if w["neuron_type"] in ("RNN", "MLP", "LiGRU", "GRU"):
        from sparch_amd.anns import ANN
        net = ANN((B, None, C), w["layer_sizes"], ann_type=w["neuron_type"], dropout=w["pdrop"],
                  normalization="batchnorm").to(dev)
    else:
        net = sparch_amd.SNN((B, None, C), w["layer_sizes"], neuron_type=w["neuron_type"], dropout=w["pdrop"],
                             normalization="batchnorm", bidirectional=bidir).to(dev)
    net.train()
    state0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}  # for the CPU baseline leg
    opt = sparch_amd.optim.Adam(net.parameters(), 1e-2)  # exp.py:89 (same arithmetic, one launch: SURVEY f-2)
    loss_fn = Fn.CrossEntropyLoss()                 # exp.py:100 (one launch for the loss and its gradient)
    reducer = dp.GradAllReducer(net, rows_per_rank=B) if multi else None
    if multi and args.sync_bn:
        Fn.SYNC_BN = {"group": None, "world": world}
    g = torch.Generator().manual_seed(4321 + rank)
    audio = w.get("audio")
    if audio:  # uniform noise in [-0.1, 0.1] + a 440 Hz tone, 1 s at 16 kHz
        tt = torch.arange(audio) / 16000.0
        x_cpu = 0.1 * (torch.rand(B, audio, generator=g) * 2 - 1) + 0.3 * torch.sin(2 * torch.pi * 440.0 * tt)[None]
    else:
        x_cpu = (torch.rand(B, T, C, generator=g) < 0.05).float()
    y_cpu = torch.randint(0, w["layer_sizes"][-1], (B,), generator=g)
    x, y = x_cpu.to(dev), y_cpu.to(dev)
    torch.manual_seed(99 + rank)

    def step():
        opt.zero_grad(set_to_none=True)
        feats = Fn.fbank(x, num_mel_bins=C) if audio else x  # nonspiking_datasets.py:96 on the device
        out, rates = net(feats)
        loss = loss_fn(out, y)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        return loss

    spiking = w["neuron_type"] not in ("RNN", "MLP", "LiGRU", "GRU")
    degraded = []  # timeouts of the persistent kernels seen by this run (all ranks agree on them)

    def kernels_ok(where):
        """Collective check of the recurrent kernels' status word.  One GPU: a timeout is an error (raise, with the
        kernel and the time step).  N > 1: every rank reports what it saw, all ranks switch to one launch per time
        step together and the caller repeats the region — a rank that raised alone used to leave its peers
        waiting in the next collective (round 2's "a rank exited")."""
        if multi:
            dp.sync_status(dev, force=world == 1)
        if not Fn.poll_status(dev):
            return True
        info = Fn.describe_timeout()
        print(f"[bench] rank {rank}: {where}: persistent-kernel timeout: {info}; status word collective over "
              f"{world} rank(s); policy {reducer.policy if reducer is not None else None}", file=sys.stderr, flush=True)
        if world == 1:
            raise sparch_amd._capi.SparchHipError(Fn._TIMEOUT_TEXT + "  [" + info + "]")
        Fn.degrade(dev)
        degraded.append(f"{where}: {info}")
        return False

    graph_mode = args.graph
    if graph_mode == "on" and not spiking:
        raise SystemExit("--graph on: spiking workloads only")
    if graph_mode == "auto":  # eager warm-up first; the host is the bound when enqueueing takes (nearly) the step
        for _ in range(max(1, args.warmup)):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        graph_mode = "on" if (spiking and not multi and enq >= 0.8 * tot) else "off"
        graph_why = f"auto: host enqueue {1e3 * enq / 3:.2f} ms of a {1e3 * tot / 3:.2f} ms step during warm-up"
    else:
        graph_why = "as requested"
        if graph_mode == "off":
            for _ in range(args.warmup):
                step()
    graphed = None
    run_step = step
    if graph_mode == "on":
        from sparch_amd.graph import GraphedTrainStep
        try:
            graphed = GraphedTrainStep(net, opt, loss_fn, x, y, reducer=reducer,
                                       front_end=(lambda a: Fn.fbank(a, num_mel_bins=C)) if audio else None,
                                       warmup=max(1, args.warmup))
            for _ in range(max(1, args.warmup)):  # untimed replays: the first launch of an instantiated graph uploads it
                graphed.step()                    # (several ms; it sat inside the 20-step timed region: cfg2 1.55 vs 1.14 ms)
            torch.cuda.synchronize()
            if os.environ.get("SPARCH_BENCH_FAIL_CAPTURE") == "1":  # exercises the fallback below
                raise RuntimeError("capture failure requested by SPARCH_BENCH_FAIL_CAPTURE")
            run_step = graphed.step
        except Exception as e:  # a failed capture must not cost the measurement: say so and launch eagerly
            if args.graph == "on":
                raise
            print(f"[bench] rank {rank}: capturing the step failed ({type(e).__name__}: {str(e)[:300]}); launching eagerly",
                  file=sys.stderr, flush=True)
            graph_why += f"; capture failed ({type(e).__name__}), eager instead"
            GraphedTrainStep.abandon(net, opt)
            graphed = None
            for _ in range(max(1, args.warmup)):
                step()
    if not kernels_ok("warm-up"):
        for _ in range(max(1, args.warmup)):  # degraded to per-step launches: warm those up
            step()
        kernels_ok("warm-up after the degrade")

    def timed_region():
        Fn.timer.reset()
        Fn.timer.enabled = graphed is None  # HIP events per named call exist only for eagerly launched kernels
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss_ = run_step()
        host_dt_ = time.perf_counter() - t0  # host time to ENQUEUE the steps (close to dt = the host is the bound)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        dt_ = time.perf_counter() - t0
        Fn.timer.enabled = False
        return dt_, host_dt_, Fn.timer.collect(), loss_

    dt, host_dt, totals, loss = timed_region()
    if not kernels_ok("timed region"):  # its steps were skipped on the device: measure the degraded path instead
        dt, host_dt, totals, loss = timed_region()
        if not kernels_ok("timed region after the degrade"):
            raise SystemExit("bench.py: persistent-kernel timeouts persist after degrading to per-step launches")
    roof_note = None
    if graphed is not None:
        # per-kernel durations for the roofline object: the same kernels launched eagerly (a graph replay has no
        # per-kernel HIP events), `steps` instrumented steps right after the timed region
        graphed.close()
        Fn.timer.reset()
        Fn.timer.enabled = True
        for _ in range(args.steps):
            step()
        Fn.timer.enabled = False
        totals = Fn.timer.collect()
        kernels_ok("instrumented eager steps")
        roof_note = (f"timed region = {args.steps} replays of the captured step; per-kernel durations from {args.steps} "
                     "eagerly launched, HIP-event instrumented steps of the same kernels right after it")
    if multi:
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
        work = algorithmic_work(dom, B * (2 if bidir else 1), T, H) if dom is not None else None
        if work is not None and work[1] is None:
            work = None
        if work is not None:
            bound, amount, _ = work
            avg_s = kern[dom]["avg_ms"] * 1e-3
            if bound == "mfma":
                ach = amount / avg_s / 1e12
                peak = PEAK_MFMA_BF16_TFLOPS if low else PEAK_MFMA_F32_TFLOPS  # bf16 mode: one bf16 MFMA per product
                traffic, traffic_src = pmc_traffic(dom, args.workload, low)
                roof = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": peak,
                        "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                        "avg_ms": kern[dom]["avg_ms"]}
                if traffic_src is not None:  # a committed profile of this workload, not a live counter read
                    roof["traffic_profile"] = traffic_src
                k6 = None if low else bf16_issue_factor(dom)
                if k6 is not None:  # the same time priced as what the kernel really issues: bf16 MFMAs of the
                    roof["frac_bf16_pipe"] = ach * k6 / PEAK_MFMA_BF16_TFLOPS  # exact split, vs the bf16 peak
                    roof["bf16_products_per_fp32_product"] = k6
            else:
                ach = amount / avg_s / 1e9
                traffic, traffic_src = pmc_traffic(dom, args.workload, low)
                roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": ach / PEAK_HBM_GBS, "traffic": traffic, "avg_ms": kern[dom]["avg_ms"]}
                if traffic_src is not None:
                    roof["traffic_profile"] = traffic_src
        print(f"[bench] gpu: {value:.0f} ts*samples/s, {ms:.2f} ms/step; dominant {dom}; "
              f"timing the CPU oracle sample next", file=sys.stderr, flush=True)
        cpu = None
        if world == 1 and not args.no_cpu_baseline and args.workload == "cfg3":
            cpu = cpu_baseline(state0, x_cpu, y_cpu)
        titles = {"cfg3": "RadLIF 3x1024 SSC shape", "cfg2": "adLIF 3x512 SHD shape",
                  "cfg4": "RadLIF 3x1024 on SC raw audio through the HIP mel front-end",
                  "cfg5": "bidirectional RadLIF 4x1024 T=1000",
                  "rnn": "RNN baseline 3x1024 SSC shape", "mlp": "MLP baseline 3x1024 SSC shape",
                  "ligru": "LiGRU baseline 3x1024 SSC shape", "gru": "GRU baseline 3x1024 SSC shape"}
        origin = (f"BASELINE.json configs[{CONFIG_INDEX[args.workload]}]" if args.workload in CONFIG_INDEX else
                  "SURVEY.md f-4: the reference's non-spiking baseline at the configs[2] shape")
        inp = (f"{audio}-sample waveforms (noise + 440 Hz tone) -> sparch_fbank_fwd -> T={T} C={C}" if audio else
               f"T={T} C={C} Bernoulli(0.05) spikes")
        note = None
        if args.workload == "ligru" and final_loss != final_loss:
            note = ("final_loss is NaN as in the reference: the real reference LiGRU (unbounded ReLU candidate, "
                    "lr 1e-2) is NaN after one Adam step at this shape too (checked in round 1)")
        line = {
            "metric": "train-step timesteps*samples/sec (fwd+bwd+Adam), " + titles[args.workload],
            "value": value, "unit": "timesteps*samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "bf16" if low else "f32", "data": "synthetic",
            "config": {"workload": f"{'bidirectional ' if bidir else ''}{w['neuron_type']} {w['layer_sizes']} batchnorm "
                                   f"pdrop={w['pdrop']}, B={B}/GPU {inp} ({origin})",
                       "global_batch": world * B, "seq_len": T, "parallelism": f"dp{world}",
                       "sync_bn": bool(world > 1 and args.sync_bn),
                       "launch": ("one captured HIP graph per step" if graphed is not None else "eager") + f" ({graph_why})",
                       "operands": ("rounded once to bf16, fp32 accumulation / states / updates" if low else
                                    "fp32 through exact bf16 splits"),
                       "grad_allreduce": (None if reducer is None else
                                          {"overlap": "all-reduce per layer, launched as its gradients appear",
                                           "window": "all-reduce per layer, in the windows between persistent launches",
                                           "deferred": "one collective after backward"}[reducer.policy])},
            "roofline": roof, "cpu_baseline": cpu,
            "kernels_ms_per_step": {k: round(v["avg_ms"] * v["launches"] / args.steps, 4) for k, v in kern.items()},
            "final_loss": final_loss,
            # host time to enqueue one step (no synchronisation inside): well below ms_per_step = the GPU is the bound
            "host_enqueue_ms_per_step": round(1e3 * host_dt / args.steps, 3),
        }
        if degraded:
            line["degraded"] = degraded  # timeouts seen; the numbers are those of the per-step-launch path
        if note:
            line["note"] = note
        if roof_note and roof is not None:
            roof["timing"] = roof_note
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
