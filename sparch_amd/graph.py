"""
The whole training step as ONE HIP graph (SURVEY.md §8 f-2 in spirit: no host work between the kernels).

A cfg3 step is ~90 kernels with 5-10 us of idle GPU between dependent launches (~0.45 ms of 8.4), a cfg2 step
is half host launch overhead.  `GraphedTrainStep` captures zero_grad -> forward -> loss -> backward ->
[gradient all-reduce] -> Adam into a `torch.cuda.CUDAGraph` (hipGraph on ROCm) and replays it.  What makes
the step capturable — every per-step VALUE that used to be a kernel argument now lives in device memory:

  * dropout seeds: a uint64 word per layer, advanced by a captured add; the kernels receive
    SPARCH_SEED_IN_MEMORY | its address (include/sparch_hip.h);
  * Adam's lr / (1 - beta1^t) and sqrt(1 - beta2^t): computed by captured ops from a device step counter
    (`optim.Adam.enable_graph_mode`);
  * the random initial states u0 / w0 / s0: static device tensors that `SNN.draw_states_into` refills from the
    CPU generator — same draws, same order as the reference — with ordinary stream-ordered copies BEFORE each
    replay (their pinned sources are fresh allocations, so a host running ahead cannot overwrite a batch the
    GPU has not consumed);
  * the batch: static `x`, `y` tensors (`load_batch`).

Nothing inside the captured region synchronises; the recurrent kernels' status word is checked by the caller
as before.  Persistent recurrent launches are ordinary kernel nodes (grid <= CU count).
"""
import torch

from . import functional as Fn


class GraphedTrainStep:
    def __init__(self, net, optimizer, loss_fn, x, y, reducer=None, extra_loss=None, front_end=None, warmup=3):
        """net: sparch_amd SNN (spiking models; the random initial states are what needs static buffers);
        x, y: example batch ON THE DEVICE (shape and dtype fixed for the graph's life);
        extra_loss(out, rates) -> tensor or None (e.g. the firing-rate regulariser, exp.py:369-372);
        front_end(x) -> features (e.g. the mel filterbank), captured with the step."""
        if not getattr(net, "is_snn", False):
            raise NotImplementedError("GraphedTrainStep: spiking networks (sparch_amd.SNN)")
        self.net, self.opt, self.loss_fn, self.reducer = net, optimizer, loss_fn, reducer
        self.extra_loss, self.front_end = extra_loss, front_end
        self.x, self.y = x.clone(), y.clone()
        dev = x.device
        # device-resident per-step values
        base = int(torch.cuda.initial_seed() & 0x3FFFFFFFFFFFFFFF)
        self._seeds = torch.tensor([base + 7919 * getattr(lay, "_layer_index", 0) for lay in net.snn],
                                   dtype=torch.int64).to(dev)  # one word per layer, advanced together (one node)
        for i, lay in enumerate(net.snn):
            lay._seed_word = self._seeds[i:i + 1]
            lay._seed_word_owner_advances = True
        optimizer.enable_graph_mode()
        feats = self.front_end(self.x) if self.front_end is not None else self.x
        self._batch = feats.shape[0]
        self._states = net.draw_states(self._batch, dev)  # becomes the static buffers
        net._static_states = self._states
        self.loss = None
        self.out = None
        self.rates = None
        # warm-up on a side stream (allocator pools, lazily initialised kernels attributes), then capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                net.draw_states_into(self._states, self._batch)
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        Fn.check_status(dev)
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        net.draw_states_into(self._states, self._batch)
        with torch.cuda.graph(self.graph):
            self._body()

    def _body(self):
        self._seeds.add_(1)
        self.opt.zero_grad(set_to_none=True)
        feats = self.front_end(self.x) if self.front_end is not None else self.x
        out, rates = self.net(feats)
        loss = self.loss_fn(out, self.y)
        if self.extra_loss is not None:
            extra = self.extra_loss(out, rates)
            if extra is not None:
                loss = loss + extra
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.opt.step()
        self.loss, self.out, self.rates = loss.detach(), out.detach(), rates.detach()

    def load_batch(self, x, y):
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)

    def step(self):
        """One training step on the batch currently in the static buffers.  Returns the (device) loss of it."""
        self.opt.sync_lr()
        self.net.draw_states_into(self._states, self._batch)
        self.graph.replay()
        self.opt.note_replay()
        return self.loss

    @staticmethod
    def abandon(net, optimizer):
        """Undo what a (possibly failed) construction left on the network and the optimizer: back to eager steps."""
        for lay in net.snn:
            lay._seed_word = None
            lay._seed_word_owner_advances = False
        net._static_states = None
        optimizer._g = None

    def close(self):
        for lay in self.net.snn:
            lay._seed_word = None
            lay._seed_word_owner_advances = False
        self.net._static_states = None
        self.opt._g = None
