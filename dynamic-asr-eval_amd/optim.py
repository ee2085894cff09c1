"""Optimisers of the adaptation step, backed by the fused HIP kernels (dyn_madgrad_step / dyn_adam_step).

Same constructor / step() / zero_grad() / state_dict() surface the reference loop uses
(`optimizer = optim(model.parameters(), **lr_args)`, `optimizer.zero_grad()`, `optimizer.step()`,
`optimizer.load_state_dict(optimizer_state)` — reference lcasr/lib.py:494-496,578-581; Adam at
nvidia_ctc/lib.py:43,155-160).  When `params` is the ParamList of our SCConformerXL the whole model is ONE flat
buffer and a step is ONE kernel launch; any other list of CUDA tensors (a foreign torch model) is stepped tensor by
tensor with the same kernel."""
import torch

from . import ops


class ParamList(list):
    """List of parameter views that also carries the flat parameter / gradient buffers they alias."""
    flat_params = None
    flat_grads = None


class _FlatOptimizer:
    def __init__(self, params, defaults):
        self.defaults = dict(defaults)
        self.k = 0
        if isinstance(params, ParamList) and params.flat_params is not None:
            self._pairs = [(params.flat_params, params.flat_grads)]
            self._foreign = None
        else:
            self._foreign = [p for p in params]
            for p in self._foreign:
                if not (isinstance(p, torch.Tensor) and p.is_cuda and p.dtype == torch.float32):
                    raise ops.DynError("optimizer: every parameter must be a float32 CUDA tensor (no CPU path)")
            self._pairs = None
        self.state = None

    def _iter_pairs(self):
        if self._pairs is not None:
            return self._pairs
        pairs = []
        for p in self._foreign:
            if getattr(p, "requires_grad", False) is False and p.grad is None:
                continue
            if p.grad is None:
                continue
            if not p.data.is_contiguous():
                raise ops.DynError("optimizer: non-contiguous parameter")
            pairs.append((p.data.view(-1), p.grad.contiguous().view(-1)))
        return pairs

    def zero_grad(self, set_to_none=False):
        if self._pairs is not None:
            for _, g in self._pairs:
                g.zero_()
        else:
            for p in self._foreign:
                p.grad = None

    def _alloc_state(self, pairs, n_bufs):
        if self.state is None:
            self.state = [[torch.empty_like(p) for _ in range(n_bufs)] for p, _ in pairs]

    def state_dict(self):
        return {"k": self.k, "defaults": dict(self.defaults),
                "state": None if self.state is None else [[b.clone() for b in bufs] for bufs in self.state]}

    def load_state_dict(self, sd):
        self.k = sd["k"]
        self.defaults.update(sd.get("defaults", {}))
        self.state = None if sd["state"] is None else [[b.clone() for b in bufs] for bufs in sd["state"]]


class MADGRAD(_FlatOptimizer):
    def __init__(self, params, lr=1e-2, momentum=0.9, weight_decay=0.0, eps=1e-6):
        if momentum < 0 or momentum >= 1:
            raise ValueError(f"Momentum {momentum} must be in the range [0,1)")
        if lr <= 0:
            raise ValueError(f"Learning rate {lr} must be positive")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, eps=eps))

    def step(self, limit=None):
        """`limit` (flat-buffer models only): step just the first `limit` elements — the replicas of a lockstep group that are still
        adapting; a finished recording's weights must not keep drifting towards its dual-averaged iterate."""
        d = self.defaults
        pairs = self._iter_pairs()
        self._alloc_state(pairs, 3)
        for (p, g), (s, nu, x0) in zip(pairs, self.state):
            if limit is not None:
                p, g, s, nu, x0 = p[:limit], g[:limit], s[:limit], nu[:limit], x0[:limit]
            ops.madgrad_step(p, g, s, nu, x0, d["lr"], d["momentum"], d["weight_decay"], d["eps"], self.k)
        self.k += 1

    def step_ranges(self, ranges):
        d = self.defaults
        _step_ranges(self, ranges, 3, lambda p, g, b, k: ops.madgrad_step(p, g, b[0], b[1], b[2], d["lr"], d["momentum"], d["weight_decay"], d["eps"], k))


def _step_ranges(opt, ranges, n_bufs, launch):
    """Shared by MADGRAD / Adam: one launch per (lo, hi, k) run of the flat buffers — the replicas of a lockstep group whose recordings have
    taken the same number of steps form one run (recordings of different lengths over several epochs, or under shuffle, get out of step: each
    keeps its OWN step count, as each would have alone)."""
    pairs = opt._iter_pairs()
    if opt._pairs is None or len(pairs) != 1:
        raise ops.DynError("step_ranges: only for a model's flat parameter buffer")
    opt._alloc_state(pairs, n_bufs)
    (p, g), bufs = pairs[0], opt.state[0]
    for lo, hi, k in ranges:
        launch(p[lo:hi], g[lo:hi], [b[lo:hi] for b in bufs], int(k))
    opt.k = max([opt.k] + [int(k) + 1 for _, _, k in ranges])


class Adam(_FlatOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    def step(self, limit=None):
        d = self.defaults
        pairs = self._iter_pairs()
        self._alloc_state(pairs, 2)
        for (p, g), (m, v) in zip(pairs, self.state):
            if limit is not None:
                p, g, m, v = p[:limit], g[:limit], m[:limit], v[:limit]
            ops.adam_step(p, g, m, v, d["lr"], d["betas"][0], d["betas"][1], d["eps"], d["weight_decay"], self.k)
        self.k += 1

    def step_ranges(self, ranges):
        d = self.defaults
        _step_ranges(self, ranges, 2, lambda p, g, b, k: ops.adam_step(p, g, b[0], b[1], d["lr"], d["betas"][0], d["betas"][1], d["eps"], d["weight_decay"], k))
