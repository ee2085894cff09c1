"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of MADGRAD, the optimiser the reference's dynamic-eval loop uses by default
(`from lcasr.optim import madgrad`, reference lcasr/lib.py:14; `optim=madgrad.MADGRAD`, lib.py:458;
`optimizer = optim(model.parameters(), **lr_args)`, lib.py:494; `optimizer.step()`, lib.py:581).

`lcasr` is an un-vendored dependency absent from /root/reference and from this container (no version is pinned
anywhere in the reference), so this follows the published algorithm of Defazio & Jelassi, "Adaptivity without
Compromise" (MADGRAD), as released in facebookresearch/madgrad: defaults lr=1e-2, momentum=0.9, weight_decay=0,
eps=1e-6; per step k:  lamb = (lr + eps) * sqrt(k + 1);  nu += lamb * g^2;  s += lamb * g;
z = x0 - s / (nu^(1/3) + eps);  p = (1 - c) p + c z with c = 1 - momentum  (x0 = p at step 0; for momentum == 0
x0 is recovered from p, s, nu before the update).  PARITY UNPINNED against lcasr.optim.madgrad: the reference holds
no fixture for it."""
import math

import torch


class MADGRAD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, momentum=0.9, weight_decay=0.0, eps=1e-6):
        if momentum < 0 or momentum >= 1:
            raise ValueError(f"Momentum {momentum} must be in the range [0,1)")
        if lr <= 0:
            raise ValueError(f"Learning rate {lr} must be positive")
        super().__init__(params, dict(lr=lr, eps=eps, momentum=momentum, weight_decay=weight_decay, k=0))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None if closure is None else closure()
        for group in self.param_groups:
            eps, k, momentum, decay = group["eps"], group["k"], group["momentum"], group["weight_decay"]
            lr = group["lr"] + eps
            ck = 1.0 - momentum
            lamb = lr * math.sqrt(k + 1)
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                st = self.state[p]
                if "grad_sum_sq" not in st:
                    st["grad_sum_sq"] = torch.zeros_like(p)
                    st["s"] = torch.zeros_like(p)
                    if momentum != 0:
                        st["x0"] = p.detach().clone()
                nu, s = st["grad_sum_sq"], st["s"]
                if decay != 0:
                    grad = grad + decay * p
                if momentum == 0:
                    rms = nu.pow(1 / 3).add_(eps)
                    x0 = p.addcdiv(s, rms, value=1)
                else:
                    x0 = st["x0"]
                nu.addcmul_(grad, grad, value=lamb)
                rms = nu.pow(1 / 3).add_(eps)
                s.add_(grad, alpha=lamb)
                if momentum == 0:
                    p.copy_(x0.addcdiv(s, rms, value=-1))
                else:
                    z = x0.addcdiv(s, rms, value=-1)
                    p.mul_(1 - ck).add_(z, alpha=ck)
            group["k"] = k + 1
        return loss
