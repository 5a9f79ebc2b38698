#!/usr/bin/env python3
"""Host-side cost of a training step (cfg2 shape by default: adLIF 3x512, B=128 — the host-bound configuration):
wall time per eagerly launched step with and without cProfile, and the profile's top entries.
usage: tools/hostprof.py [cfg3] [profile]   (run from the tree to measure: `cd .ab/r2 && python ../../tools/hostprof.py`)"""
import cProfile
import io
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
import sparch_amd  # noqa: E402
from sparch_amd import functional as Fn  # noqa: E402

dev = torch.device("cuda", 0)
cfg3 = "cfg3" in sys.argv
B, T, C = (256, 250, 700) if cfg3 else (128, 250, 700)
torch.manual_seed(1234)
net = sparch_amd.SNN((B, None, C), [1024, 1024, 35] if cfg3 else [512, 512, 20], neuron_type="RadLIF" if cfg3 else "adLIF",
                     dropout=0.1, normalization="batchnorm").to(dev).train()
opt = sparch_amd.optim.Adam(net.parameters(), 1e-2)
loss_fn = getattr(Fn, "CrossEntropyLoss", torch.nn.CrossEntropyLoss)()
g = torch.Generator().manual_seed(4321)
x = (torch.rand(B, T, C, generator=g) < 0.05).float().to(dev)
y = torch.randint(0, 35 if cfg3 else 20, (B,), generator=g).to(dev)


def step():
    opt.zero_grad(set_to_none=True)
    out, rates = net(x)
    loss = loss_fn(out, y)
    loss.backward()
    opt.step()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
enq = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"eager: host enqueue {1e3 * enq / 200:.3f} ms per step, step {1e3 * tot / 200:.3f} ms")
if "profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
    print(s.getvalue()[:5000])

if "graph" in sys.argv:  # host cost of the parts of a graph-replayed step
    from sparch_amd.graph import GraphedTrainStep
    g = GraphedTrainStep(net, opt, loss_fn, x, y)
    if "first" in sys.argv:  # the first replays after the capture, one by one (host ms: draws, replay)
        torch.cuda.synchronize()
        rows = []
        for i in range(40):
            t0 = time.perf_counter(); g.opt.sync_lr(); g.net.draw_states_into(g._states, g._batch)
            t1 = time.perf_counter(); g.graph.replay()
            t2 = time.perf_counter(); g.opt.note_replay()
            rows.append((round(1e3 * (t1 - t0), 3), round(1e3 * (t2 - t1), 3)))
        torch.cuda.synchronize()
        print("first replays (draw ms, replay ms):", rows)
    for _ in range(10):
        g.step()
    torch.cuda.synchronize()
    parts = {"sync_lr": 0.0, "draw_states": 0.0, "replay": 0.0, "note": 0.0}
    n = 200
    t_all = time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter(); g.opt.sync_lr()
        t1 = time.perf_counter(); g.net.draw_states_into(g._states, g._batch)
        t2 = time.perf_counter(); g.graph.replay()
        t3 = time.perf_counter(); g.opt.note_replay()
        t4 = time.perf_counter()
        parts["sync_lr"] += t1 - t0; parts["draw_states"] += t2 - t1; parts["replay"] += t3 - t2; parts["note"] += t4 - t3
    enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    tot = time.perf_counter() - t_all
    print(f"graph: host {1e3 * enq / n:.3f} ms per step, step {1e3 * tot / n:.3f} ms; parts (ms): "
          + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in parts.items()))
    # the same with the GPU idle in between (pure host cost of a replay)
    r = 0.0
    for _ in range(50):
        g.opt.sync_lr(); g.net.draw_states_into(g._states, g._batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); g.graph.replay(); r += time.perf_counter() - t0
        torch.cuda.synchronize()
    print(f"graph.replay() on an idle queue: {1e3 * r / 50:.3f} ms")
    for rep in range(6):  # windows of 20 replayed steps, a synchronize before and after each (as bench.py's timed region)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.step()
        enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"window {rep}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per step (host {1e3 * enq / 20:.3f})")
