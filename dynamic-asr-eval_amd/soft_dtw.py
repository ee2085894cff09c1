"""SoftDTW with the reference's module interface — `SoftDTW(use_cuda, gamma=1.0, normalize=False, bandwidth=None,
dist_func=None)` and `forward(X, Y) -> [B]` (reference wav2vec2/soft_dtw_cuda.py:273-352; constructed at
wav2vec2/lib.py:130,370 as `SoftDTW(use_cuda=True, gamma=1.5)`) — on the HIP kernels dyn_sqdist / dyn_softdtw_fwd /
dyn_softdtw_bwd.  Differentiable w.r.t. X through a torch.autograd.Function (autograd is only the plumbing that hands
grad_output to the backward kernels, as in the reference's `_SoftDTWCUDA`).  Sequence lengths are not limited to 1024."""
import torch

from . import ops
from ._lib import check


def _L():
    from . import _lib
    return _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def sqdist(x, y):
    x = ops._c(x.contiguous(), "sqdist.x"); y = ops._c(y.contiguous(), "sqdist.y")
    B, N, d = x.shape
    M = y.shape[1]
    D = torch.empty(B, N, M, device=x.device, dtype=torch.float32)
    check(_L().dyn_sqdist(x.data_ptr(), y.data_ptr(), D.data_ptr(), B, N, M, d, _stream()), "dyn_sqdist")
    return D


def softdtw_forward(D, gamma, bandwidth=0.0):
    D = ops._c(D, "softdtw.D")
    B, N, M = D.shape
    R = torch.empty(B, N + 2, M + 2, device=D.device, dtype=torch.float64)  # fp64 lattice workspace
    value = torch.empty(B, device=D.device, dtype=torch.float32)
    check(_L().dyn_softdtw_fwd(D.data_ptr(), R.data_ptr(), value.data_ptr(), B, N, M, gamma, bandwidth, _stream()), "dyn_softdtw_fwd")
    return value, R


def softdtw_backward(D, R, gamma, bandwidth=0.0):
    B, N, M = D.shape
    E = torch.empty(B, N, M, device=D.device, dtype=torch.float32)
    check(_L().dyn_softdtw_bwd(D.data_ptr(), R.data_ptr(), E.data_ptr(), B, N, M, gamma, bandwidth, _stream()), "dyn_softdtw_bwd")
    return E


class _SoftDTWFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Y, gamma, bandwidth):
        Xc, Yc = X.detach().contiguous(), Y.detach().contiguous()
        D = sqdist(Xc, Yc)
        value, R = softdtw_forward(D, gamma, bandwidth)
        ctx.save_for_backward(Xc, Yc, D, R)
        ctx.gamma, ctx.bandwidth = gamma, bandwidth
        return value

    @staticmethod
    def backward(ctx, grad_output):
        Xc, Yc, D, R = ctx.saved_tensors
        E = softdtw_backward(D, R, ctx.gamma, ctx.bandwidth)
        G = (grad_output.reshape(-1, 1, 1).to(torch.float32) * E).contiguous()   # reference :174 (broadcast multiply)
        B, N, d = Xc.shape
        M = Yc.shape[1]
        dX = torch.empty_like(Xc)
        check(_L().dyn_sqdist_bwd_x(Xc.data_ptr(), Yc.data_ptr(), G.data_ptr(), dX.data_ptr(), B, N, M, d, _stream()), "dyn_sqdist_bwd_x")
        dY = None
        if ctx.needs_input_grad[1]:
            Gt = G.transpose(1, 2).contiguous()
            dY = torch.empty_like(Yc)
            check(_L().dyn_sqdist_bwd_x(Yc.data_ptr(), Xc.data_ptr(), Gt.data_ptr(), dY.data_ptr(), B, M, N, d, _stream()), "dyn_sqdist_bwd_x")
        return dX, dY, None, None


class SoftDTW(torch.nn.Module):
    def __init__(self, use_cuda=True, gamma=1.0, normalize=False, bandwidth=None, dist_func=None):
        super().__init__()
        if not use_cuda:
            raise ops.DynError("SoftDTW: only the GPU (HIP) implementation exists in this package")
        if dist_func is not None:
            raise NotImplementedError("custom dist_func: only the squared Euclidean distance is fused")
        self.normalize, self.gamma = normalize, float(gamma)
        self.bandwidth = 0.0 if bandwidth is None else float(bandwidth)

    def forward(self, X, Y):
        bx, lx, dx = X.shape
        by, ly, dy = Y.shape
        assert bx == by and dx == dy
        if self.normalize:  # reference :342-349
            x = torch.cat([X, X, Y]); y = torch.cat([Y, X, Y])
            out = _SoftDTWFn.apply(x, y, self.gamma, self.bandwidth)
            out_xy, out_xx, out_yy = torch.split(out, X.shape[0])
            return out_xy - 1 / 2 * (out_xx + out_yy)
        return _SoftDTWFn.apply(X, Y, self.gamma, self.bandwidth)
