"""
Optimizer step on the device in one launch.

`Adam` is a drop-in for the reference's `torch.optim.Adam(net.parameters(), lr)` (exp.py:89): same
constructor arguments, same `param_groups` (so `ReduceLROnPlateau`, exp.py:91-98, drives `lr` unchanged),
same `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter) — a checkpoint of one loads into
the other.  The arithmetic is torch's default Adam path operation by operation (csrc/optim.hip); `step`
needs no host round trip.  amsgrad / maximize / capturable variants are not part of the reference's use and
raise.
"""
import ctypes
import math

import torch

from ._capi import check, lib
from .functional import _stream, status_word


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        if amsgrad:
            raise NotImplementedError("sparch_amd.optim.Adam: amsgrad is not implemented (the reference does not use it)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        # Steps the device skipped after a persistent kernel's timeout (status word raised: sparch_adam_step is a
        # no-op and counts itself in status[1]) are taken back from the bias-correction counters when the host
        # reads the word, so that "a skipped step is a no-op" also holds for t in 1 - beta^t.
        import weakref

        from . import functional as _Fn
        ref = weakref.ref(self)

        def _on_timeout(info, ref=ref):
            opt = ref()
            if opt is None:
                _Fn._timeout_listeners.remove(_on_timeout)
            else:
                opt.take_back(info["skipped_steps"])
        _Fn._timeout_listeners.append(_on_timeout)

    def take_back(self, n):
        """Undo the step-counter advance of `n` optimizer steps that were no-ops on the device."""
        if n <= 0:
            return
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st and "step" in st:
                    st["step"] -= min(float(n), float(st["step"]))
        g = getattr(self, "_g", None)
        if g is not None:
            g["t"].sub_(float(n)).clamp_(min=0.0)

    def enable_graph_mode(self):
        """Keep the per-step factors (lr / (1 - beta1^t), sqrt(1 - beta2^t)) in DEVICE memory, computed by a few
        captured torch ops from a device step counter, so that `step()` can be part of a HIP graph (kernel
        arguments are frozen at capture time).  The arithmetic is the same double-precision formula the host
        path evaluates; `lr` is read from a device scalar that `step()` refreshes whenever the scheduler changed
        the group's value.  One parameter group stepping together (the reference's use)."""
        if len(self.param_groups) != 1:
            raise NotImplementedError("sparch_amd.optim.Adam graph mode: one parameter group")
        group = self.param_groups[0]
        dev = group["params"][0].device
        t0 = 0.0
        for p in group["params"]:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            t0 = float(st["step"])
        self._g = {"t": torch.full((), t0, dtype=torch.float64, device=dev),
                   "lr": torch.full((), float(group["lr"]), dtype=torch.float64, device=dev),
                   "lr_host": float(group["lr"]),
                   "scalars": torch.zeros(2, dtype=torch.float32, device=dev)}

    def sync_lr(self):
        """Graph mode, outside the captured region: push a learning rate changed by the scheduler."""
        g = getattr(self, "_g", None)
        if g is not None and float(self.param_groups[0]["lr"]) != g["lr_host"]:
            g["lr_host"] = float(self.param_groups[0]["lr"])
            g["lr"].fill_(g["lr_host"])

    def note_replay(self):
        """Graph mode: a replay advanced the device step counter; keep the host-side `step` entries in step."""
        for p in self.param_groups[0]["params"]:
            if p in self.state:
                self.state[p]["step"] += 1

    @torch.no_grad()
    def step(self, closure=None):
        if getattr(self, "_g", None) is not None:
            return self._step_graph_mode()
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            beta1, beta2 = group["betas"]
            steps = set()
            for p in ps:
                if p.dtype != torch.float32 or not p.is_cuda or p.grad.is_sparse:
                    raise RuntimeError("sparch_amd.optim.Adam: float32 dense parameters on the GPU only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)  # host scalar, as torch.optim.Adam keeps it
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                steps.add(float(st["step"]))
            # parameters that joined at different times step separately (never the case in the reference's use)
            for t in sorted(steps):
                sel = [p for p in ps if float(self.state[p]["step"]) == t]
                bc1 = 1.0 - beta1 ** t
                bc2 = 1.0 - beta2 ** t
                self._launch(sel, group["lr"] / bc1, beta1, beta2, math.sqrt(bc2), group["eps"], group["weight_decay"])
        return loss

    def _step_graph_mode(self):
        group, g = self.param_groups[0], self._g
        ps = [p for p in group["params"] if p.grad is not None]
        beta1, beta2 = group["betas"]
        check(lib.sparch_adam_scalars(g["t"].data_ptr(), g["lr"].data_ptr(), float(beta1), float(beta2),
                                      g["scalars"].data_ptr(), _stream()), "sparch_adam_scalars")
        if not torch.cuda.is_current_stream_capturing():
            self.note_replay()
        self._launch(ps, 0.0, beta1, beta2, 1.0, group["eps"], group["weight_decay"], scalars=g["scalars"])
        return None

    def _launch(self, ps, step_size, beta1, beta2, bc2_sqrt, eps, weight_decay, scalars=None):
        n = len(ps)
        arr = ctypes.c_void_p * n
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
        for p in ps:
            if not p.is_contiguous():
                raise RuntimeError("sparch_amd.optim.Adam: parameters must be contiguous")
        a_p = arr(*[p.data_ptr() for p in ps])
        a_g = arr(*[g.data_ptr() for g in grads])
        a_m = arr(*[self.state[p]["exp_avg"].data_ptr() for p in ps])
        a_v = arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in ps])
        a_n = (ctypes.c_int64 * n)(*[p.numel() for p in ps])
        # guarded by the recurrent kernels' status word: after an in-kernel timeout (invalid gradients) the
        # step is a no-op on the device until check_status() has reported it and cleared the word
        skip = status_word(ps[0].device)
        check(lib.sparch_adam_step(n, a_p, a_g, a_m, a_v, a_n, float(step_size), float(beta1), float(beta2),
                                   float(bc2_sqrt), float(eps), float(weight_decay),
                                   scalars.data_ptr() if scalars is not None else None, skip.data_ptr(), _stream()),
              "sparch_adam_step")
