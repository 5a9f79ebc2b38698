"""
Single-node data parallelism for the spiking training step: one process per GPU,
gradient all-reduce (mean) over RCCL/xGMI.  NEW functionality — the reference is
single-device only (exp.py:81 is its one device decision).

Samples are independent through the whole time loop, so the batch shards across
ranks with no data-path exchange; the only collective is the gradient all-reduce:
15.6 MB fp32 at RadLIF 3x1024.  A layer's backward is ONE autograd node (its reverse
time loop + GEMMs), so all of a layer's parameter gradients appear together.  We
bucket per layer: a post-accumulate-grad hook counts a layer's parameters and, when
the last one lands, flattens the bucket.  Two launch policies:

  * overlap (models WITHOUT recurrent layers): the bucket's all-reduce is launched at once, async, and
    runs on RCCL's stream underneath the next (earlier) layer's backward;
  * deferred (models whose layers run the PERSISTENT recurrent kernels — RLIF / RadLIF / RNN / LiGRU / GRU up
    to 1024 hidden units; keyed on the layers' `uses_persistent_kernel`, not on having a V matrix: the
    large-H step path launches per time step and overlaps like everything else — AND whose persistent grid
    needs the whole GPU): ONE all-reduce of all buckets at the end of backward.  Those kernels' workgroups
    wait for each other and need one CU each: an RCCL kernel that holds a few CUs while it waits for a peer
    rank keeps a 256-workgroup grid from becoming co-resident, and a peer whose own persistent kernel got
    the GPU first keeps ITS RCCL kernel from starting — a rank that is slightly ahead would then lose up to
    a whole recurrent launch (1.3 ms) per bucket.  The deferred collective costs a fixed ~0.1-0.2 ms
    (15.6 MB over 7 xGMI links) instead.  When the per-rank batch leaves CUs free (`rows_per_rank` given:
    ceil(rows/32) row tiles x H/32 column tiles <= CUs - RCCL_CU_RESERVE, e.g. any strong-scaling shard of
    the headline batch), the grid and RCCL's channels fit side by side and the overlap policy is used.
    `SPARCH_DP_OVERLAP=0/1` overrides.

`finish()` waits, averages and hands the gradients to `optimizer.step()`.

BatchNorm statistics stay per-rank (standard DDP semantics).
Works with any torch.distributed backend: "nccl" (= RCCL) on GPUs, "gloo" in CPU tests.
"""
import torch
import torch.distributed as dist


RCCL_CU_RESERVE = 32  # CUs left to RCCL's channel kernels when a persistent grid runs beside them


def persistent_grid_fills_gpu(layers, rows_per_rank, cus=256):
    """True if some layer's persistent recurrent launch would occupy (nearly) every CU: n_row_tiles x
    n_column_tiles workgroups, one per CU, against cus - RCCL_CU_RESERVE.  Unknown batch -> assume it does."""
    for lay in layers:
        if not getattr(lay, "uses_persistent_kernel", False):
            continue
        if rows_per_rank is None:
            return True
        rows = rows_per_rank * (2 if getattr(lay, "bidirectional", False) else 1)
        per_wg = int(getattr(lay, "persistent_units_per_workgroup", 32))  # hidden units (columns) per workgroup
        n_rt, n_ct = -(-rows // 32), -(-int(lay.hidden_size) // per_wg)
        if n_ct * min(n_rt, max(1, cus // n_ct)) > cus - RCCL_CU_RESERVE:
            return True
    return False


class GradAllReducer:
    def __init__(self, module, process_group=None, buckets=None, overlap=None, rows_per_rank=None):
        """buckets: list of lists of parameters (default: one bucket per child of `module.snn` / `module.ann`,
        or a single bucket for arbitrary modules).  Buckets fire in whatever order backward
        completes them (readout first, input layer last).  overlap: None = by model and per-rank batch
        (`rows_per_rank`), see the module docstring."""
        import os

        self.group = process_group
        layers_all = getattr(module, "snn", None) or getattr(module, "ann", None) or []
        if overlap is None:
            env = os.environ.get("SPARCH_DP_OVERLAP", "")
            if env in ("0", "1"):
                overlap = env == "1"
            else:
                cus = 256
                if torch.cuda.is_available():
                    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
                overlap = not persistent_grid_fills_gpu(layers_all, rows_per_rank, cus)
        self.overlap = bool(overlap)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if buckets is None:
            layers = getattr(module, "snn", None) or getattr(module, "ann", None)  # SNN / ANN layer lists
            if layers is not None:
                buckets = [[p for p in layer.parameters() if p.requires_grad] for layer in layers]
            else:
                buckets = [[p for p in module.parameters() if p.requires_grad]]
        self.buckets = [b for b in buckets if b]
        self._pending = [0] * len(self.buckets)
        self._flat = [None] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._handles = []
        self.bytes_per_step = sum(p.numel() * 4 for b in self.buckets for p in b)
        for bi, bucket in enumerate(self.buckets):
            for p in bucket:
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self.reset()

    def reset(self):
        for bi, b in enumerate(self.buckets):
            self._pending[bi] = len(b)
            self._flat[bi] = None
            self._work[bi] = None

    def _make_hook(self, bi):
        def hook(_param):
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi)
        return hook

    def _launch(self, bi):
        if self.world == 1 or not self.overlap:
            return
        grads = [p.grad for p in self.buckets[bi]]
        flat = torch.cat([g.reshape(-1) for g in grads])
        self._flat[bi] = flat
        self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Wait for every bucket, turn the sums into means and hand them to the optimizer, re-arm for the
        next step.  The averaged gradients stay in the flat bucket: each `.grad` becomes a view of it (one
        scaling kernel per bucket, no copy back)."""
        if self.world > 1 and not self.overlap:
            for bi, bucket in enumerate(self.buckets):
                if self._pending[bi] not in (0, len(bucket)):
                    raise RuntimeError("GradAllReducer: a bucket received only part of its gradients")
            ps = [p for bi, bucket in enumerate(self.buckets) if self._pending[bi] == 0 for p in bucket]
            if ps:
                flat = torch.cat([p.grad.reshape(-1) for p in ps])
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                flat.mul_(1.0 / self.world)
                off = 0
                for p in ps:
                    n = p.numel()
                    p.grad = flat[off:off + n].view_as(p)
                    off += n
        elif self.world > 1:
            inv = 1.0 / self.world
            for bi, bucket in enumerate(self.buckets):
                if self._work[bi] is None:
                    if self._pending[bi] != len(bucket) and self._pending[bi] != 0:
                        raise RuntimeError("GradAllReducer: a bucket received only part of its gradients")
                    if self._pending[bi] == len(bucket):
                        continue  # layer took no part in this backward
                self._work[bi].wait()
                flat = self._flat[bi]
                flat.mul_(inv)
                off = 0
                for p in bucket:
                    n = p.numel()
                    p.grad = flat[off:off + n].view_as(p)
                    off += n
        self.reset()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


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
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
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
