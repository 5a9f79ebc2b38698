"""
Single-node data parallelism for the spiking training step: one process per GPU,
gradient all-reduce (mean) over RCCL/xGMI.  NEW functionality — the reference is
single-device only (exp.py:81 is its one device decision).

Samples are independent through the whole time loop, so the batch shards across
ranks with no data-path exchange; the only collective is the gradient all-reduce:
15.6 MB fp32 at RadLIF 3x1024.  A layer's backward is ONE autograd node (its reverse
time loop + GEMMs), so all of a layer's parameter gradients appear together.  We
bucket per layer: a post-accumulate-grad hook counts a layer's parameters and, when
the last one lands, flattens the bucket.  Three launch policies:

  * "overlap" (models WITHOUT layers on the persistent recurrent kernels): the bucket's all-reduce is launched
    at once, async, and runs on RCCL's stream underneath the next (earlier) layer's backward;
  * "window" (round 3; the default for models whose layers run the PERSISTENT recurrent kernels — RLIF / RadLIF
    / RNN / LiGRU / GRU up to 1024 hidden units; keyed on the layers' `uses_persistent_kernel`): those kernels'
    workgroups wait for each other and need a CU each, so a collective must never be resident beside one —
    an RCCL kernel that holds a few CUs while it waits for a peer rank keeps a 256-workgroup grid from
    becoming co-resident.  But a layer's backward is a persistent launch FOLLOWED by ~1.3 ms of ordinary
    GEMMs (dV, the BatchNorm pass, dW, dX) that share CUs with anything.  The rule: a ready bucket is launched
    right BEHIND the next persistent launch (`functional` calls `post_persistent()`; the collective's stream
    waits for everything enqueued so far, i.e. it starts when that kernel has finished and runs under the
    GEMMs that follow), and the compute stream waits for every collective in flight right BEFORE the next
    persistent launch (`pre_persistent()`).  The last bucket (the input layer's: no persistent launch
    follows) goes out in `finish()`.  No collective is ever enqueued between the start and the end of a
    persistent kernel, and none is in flight when one starts.  NOT measured on RCCL (the build box has one GPU):
    correctness and the event order are covered with gloo;
  * "deferred": ONE all-reduce of all buckets at the end of backward (`SPARCH_DP_OVERLAP=0`).

`SPARCH_DP_POLICY=overlap|window|deferred` (or the older `SPARCH_DP_OVERLAP=0/1` = deferred / overlap) overrides.
Rounds 1-2 chose "overlap" for persistent grids that leave >= 32 CUs free (strong-scaling shards); that reserve
was a guess about RCCL's channel kernels and is gone: such models take "window" at any batch size.

`finish()` waits, averages, hands the gradients to `optimizer.step()` — and makes the recurrent kernels' STATUS
WORD collective (MAX over ranks, on the device): after an in-kernel timeout on ONE rank that rank's gradients are
invalid, they have been averaged into every peer, so every rank must skip the same optimizer steps (the device
skip word of sparch_adam_step / sparch_bn_finalize) and later degrade to per-step launches together.

BatchNorm statistics stay per-rank (standard DDP semantics).
Works with any torch.distributed backend: "nccl" (= RCCL) on GPUs, "gloo" in CPU tests.
"""
import os

import torch
import torch.distributed as dist

POLICIES = ("overlap", "window", "deferred")


def has_persistent_layers(layers):
    return any(getattr(lay, "uses_persistent_kernel", False) for lay in layers)


def persistent_grid_fills_gpu(layers, rows_per_rank, cus=256, reserve=32):
    """True if some layer's persistent recurrent launch would occupy (nearly) every CU: n_row_tiles x
    n_column_tiles workgroups, one per CU, against cus - reserve.  Unknown batch -> assume it does.
    (Informational since round 3: the policy no longer depends on it.)"""
    for lay in layers:
        if not getattr(lay, "uses_persistent_kernel", False):
            continue
        if rows_per_rank is None:
            return True
        rows = rows_per_rank * (2 if getattr(lay, "bidirectional", False) else 1)
        per_wg = int(getattr(lay, "persistent_units_per_workgroup", 32))  # hidden units (columns) per workgroup
        n_rt, n_ct = -(-rows // 32), -(-int(lay.hidden_size) // per_wg)
        if n_ct * min(n_rt, max(1, cus // n_ct)) > cus - reserve:
            return True
    return False


def choose_policy(layers):
    env = os.environ.get("SPARCH_DP_POLICY", "")
    if env in POLICIES:
        return env
    legacy = os.environ.get("SPARCH_DP_OVERLAP", "")
    if legacy in ("0", "1"):
        return "overlap" if legacy == "1" else "deferred"
    return "window" if has_persistent_layers(layers) else "overlap"


def sync_status(device, group=None, force=False):
    """Make the recurrent kernels' status word the same on every rank (MAX of the raised flag, on the device, no
    host synchronisation).  Call before anything that branches on it — `functional.check_status` at an epoch
    end, the bench's checks — so that all ranks degrade or raise together instead of one leaving its peers in a
    collective."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    from . import functional as Fn

    word = Fn.status_word(device)
    dist.all_reduce(word[:1], op=dist.ReduceOp.MAX, group=group)


class GradAllReducer:
    def __init__(self, module, process_group=None, buckets=None, overlap=None, rows_per_rank=None, policy=None,
                 status_device=None, trace=False):
        """buckets: list of lists of parameters (default: one bucket per child of `module.snn` / `module.ann`,
        or a single bucket for arbitrary modules).  Buckets fire in whatever order backward
        completes them (readout first, input layer last).  policy: None = by model (module docstring);
        overlap (older interface): True / False = "overlap" / "deferred".  rows_per_rank: informational.
        status_device: device whose status word `finish()` makes collective (default: the parameters' device when
        it is a GPU).  trace: keep an ordered log of (event, index) in `self.trace` (tests)."""
        self.group = process_group
        layers_all = getattr(module, "snn", None) or getattr(module, "ann", None) or []
        if policy is None and overlap is not None:
            policy = "overlap" if overlap else "deferred"
        if policy is None:
            policy = choose_policy(layers_all)
        if policy not in POLICIES:
            raise ValueError(f"GradAllReducer: policy must be one of {POLICIES}, got {policy!r}")
        self.policy = policy
        self.overlap = policy != "deferred"  # some collective runs under the backward pass
        self.rows_per_rank = rows_per_rank
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # collectives are issued when there is a peer — or when SPARCH_DP_FORCE_COLLECTIVES=1 asks for them in a
        # one-rank group (tools/nccl_world1_check.py: the only way to run the RCCL calls, their stream semantics and
        # their coexistence with the persistent kernels on a one-GPU box; a sum over one rank is the identity)
        self._collective = self.world > 1 or (dist.is_initialized()
                                              and os.environ.get("SPARCH_DP_FORCE_COLLECTIVES", "0") == "1")
        if buckets is None:
            layers = getattr(module, "snn", None) or getattr(module, "ann", None)  # SNN / ANN layer lists
            if layers is not None:
                buckets = [[p for p in layer.parameters() if p.requires_grad] for layer in layers]
            else:
                buckets = [[p for p in module.parameters() if p.requires_grad]]
        self.buckets = [b for b in buckets if b]
        if status_device is None:
            p0 = self.buckets[0][0] if self.buckets else None
            status_device = p0.device if (p0 is not None and p0.is_cuda) else None
        self.status_device = status_device
        self._pending = [0] * len(self.buckets)
        self._flat = [None] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._ready = []      # window policy: buckets flattened, waiting for the next persistent launch to pass
        self._inflight = []   # window policy: launched, not yet waited for on the compute stream
        self._handles = []
        self.trace = [] if trace else None
        self.bytes_per_step = sum(p.numel() * 4 for b in self.buckets for p in b)
        for bi, bucket in enumerate(self.buckets):
            for p in bucket:
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        if self.policy == "window":
            from . import functional as Fn
            Fn.persistent_hooks.append(self)
        self.reset()

    def _log(self, what, i):
        if self.trace is not None:
            self.trace.append((what, i))

    def reset(self):
        for bi, b in enumerate(self.buckets):
            self._pending[bi] = len(b)
            self._flat[bi] = None
            self._work[bi] = None
        self._ready, self._inflight = [], []

    def _make_hook(self, bi):
        def hook(_param):
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._bucket_complete(bi)
        return hook

    def _flatten(self, bi):
        self._flat[bi] = torch.cat([p.grad.reshape(-1) for p in self.buckets[bi]])

    def _launch(self, bi):
        self._work[bi] = dist.all_reduce(self._flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._log("launch", bi)

    def _bucket_complete(self, bi):
        self._log("ready", bi)
        if not self._collective or self.policy == "deferred":
            return
        self._flatten(bi)
        if self.policy == "overlap":
            self._launch(bi)
        else:  # window: behind the next persistent launch (post_persistent) or in finish()
            self._ready.append(bi)

    # ---- window policy: called by sparch_amd.functional around every persistent recurrent launch
    def pre_persistent(self):
        """Right before a persistent launch is enqueued: nothing of ours may be in flight beside it — the compute
        stream waits for every collective launched so far."""
        self._log("pre", len(self._inflight))
        for bi in self._inflight:
            self._work[bi].wait()
            self._log("wait", bi)
        self._inflight = []

    def post_persistent(self):
        """Right behind a persistent launch: the buckets that became ready before it go out now; their
        collective starts when that kernel has finished (stream order) and runs under the GEMMs that follow."""
        self._log("post", len(self._ready))
        if not self._collective:
            return
        for bi in self._ready:
            self._launch(bi)
            self._inflight.append(bi)
        self._ready = []

    def finish(self):
        """Wait for every bucket, turn the sums into means and hand them to the optimizer, make the status word
        collective, re-arm for the next step.  The averaged gradients stay in the flat bucket: each `.grad`
        becomes a view of it (one scaling kernel per bucket, no copy back)."""
        for bi, bucket in enumerate(self.buckets):
            if self._pending[bi] not in (0, len(bucket)):
                raise RuntimeError("GradAllReducer: a bucket received only part of its gradients")
        if self._collective and self.policy == "deferred":
            ps = [p for bi, bucket in enumerate(self.buckets) if self._pending[bi] == 0 for p in bucket]
            if ps:
                flat = torch.cat([p.grad.reshape(-1) for p in ps])
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                self._log("launch", -1)
                flat.mul_(1.0 / self.world)
                off = 0
                for p in ps:
                    n = p.numel()
                    p.grad = flat[off:off + n].view_as(p)
                    off += n
        elif self._collective:
            for bi in self._ready:  # window: what no persistent launch followed (the input layer's bucket)
                self._launch(bi)
            self._ready, self._inflight = [], []
            inv = 1.0 / self.world
            for bi, bucket in enumerate(self.buckets):
                if self._work[bi] is None:
                    continue  # layer took no part in this backward
                self._work[bi].wait()
                self._log("wait", bi)
                flat = self._flat[bi]
                flat.mul_(inv)
                off = 0
                for p in bucket:
                    n = p.numel()
                    p.grad = flat[off:off + n].view_as(p)
                    off += n
        if self._collective and self.status_device is not None:
            sync_status(self.status_device, self.group, force=self.world == 1)
        self.reset()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        from . import functional as Fn
        if self in Fn.persistent_hooks:
            Fn.persistent_hooks.remove(self)


def shard_batch(x, rank, world):
    """Rows [rank*B/world, (rank+1)*B/world) of a global batch (SURVEY.md §8e partitioning)."""
    B = x.shape[0]
    if B % world != 0:
        raise ValueError(f"global batch {B} is not divisible by world size {world}")
    per = B // world
    return x[rank * per:(rank + 1) * per]


def init_from_env(backend=None):
    """torch.distributed rendezvous from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    force = os.environ.get("SPARCH_DP_FORCE_COLLECTIVES", "0") == "1" and "RANK" in os.environ  # one-rank rehearsal on RCCL
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:  # SPARCH_DIST_BACKEND=gloo: rehearsal of the multi-process path without RCCL
            backend = os.environ.get("SPARCH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("SPARCH_SHARE_GPU", "0") == "1":
            local = 0  # rehearsal on a one-GPU box: every rank on cuda:0
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
