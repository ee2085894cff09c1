"""Thin tensor-level wrappers over the C-ABI (include/dyneval.h).

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic op on the product path is a
hand-written HIP kernel reached through ctypes.  Wrappers validate device/dtype/contiguity on the host so a
kernel never sees a shape it does not expect."""
import ctypes

import torch

from . import _lib
from ._lib import GemmDesc, check

_WS = {}
WORKSPACE_BYTES = 768 << 20


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.DynError(f"{name}: expected a CUDA tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.DynError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t


def _cc(t, name, dtype=torch.float32):
    _chk(t, name, dtype)
    if not t.is_contiguous():
        raise _lib.DynError(f"{name}: expected a contiguous tensor")
    return t


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def workspace(device=None):
    """One caller-owned scratch buffer per device (split-K slabs, partial reductions)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
        _WS[key] = ws
    return ws


def gemm(a, b, c, *, trans_a=False, trans_b=False, M, N, K, lda, ldb, ldc, alpha=1.0, beta=0.0, bias=None,
         nb1=1, nb2=1, sa=(0, 0), sb=(0, 0), sc=(0, 0), a_off=0, b_off=0, c_off=0, split_k=0):
    """Raw strided batched GEMM (see dyn_gemm_desc). Offsets are in elements from each tensor's data_ptr."""
    _chk(a, "gemm.A"); _chk(b, "gemm.B"); _chk(c, "gemm.C")
    d = GemmDesc()
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    d.M, d.N, d.K = M, N, K
    d.alpha, d.beta = alpha, beta
    d.A, d.lda, d.sa1, d.sa2 = a.data_ptr() + 4 * a_off, lda, sa[0], sa[1]
    d.B, d.ldb, d.sb1, d.sb2 = b.data_ptr() + 4 * b_off, ldb, sb[0], sb[1]
    d.C, d.ldc, d.sc1, d.sc2 = c.data_ptr() + 4 * c_off, ldc, sc[0], sc[1]
    d.bias = bias.data_ptr() if bias is not None else None
    d.nb1, d.nb2 = nb1, nb2
    d.split_k = split_k
    ws = workspace(c.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    check(_lib.load().dyn_gemm_f32(ctypes.byref(d), _stream()), "dyn_gemm_f32")
    return c


def linear(x, w, bias=None, out=None):
    """y[M, N] = x[M, K] @ w[N, K]^T + bias  (torch.nn.Linear layout)."""
    _cc(x, "linear.x"); _cc(w, "linear.w")
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    return gemm(x, w, out, trans_b=True, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias)


def linear_dgrad(dy, w, out=None, beta=0.0):
    """dx[M, K] = dy[M, N] @ w[N, K]."""
    _cc(dy, "linear_dgrad.dy"); _cc(w, "linear_dgrad.w")
    N, K = w.shape
    M = dy.numel() // N
    if out is None:
        out = torch.empty(*dy.shape[:-1], K, device=dy.device, dtype=torch.float32)
    return gemm(dy, w, out, M=M, N=K, K=N, lda=N, ldb=K, ldc=K, beta=beta)


def linear_wgrad(dy, x, dw, beta=1.0):
    """dw[N, K] (+)= dy[M, N]^T @ x[M, K]  (deterministic split-K)."""
    _cc(dy, "linear_wgrad.dy"); _cc(x, "linear_wgrad.x"); _cc(dw, "linear_wgrad.dw")
    N, K = dw.shape
    M = dy.numel() // N
    return gemm(dy, x, dw, trans_a=True, M=N, N=K, K=M, lda=N, ldb=K, ldc=K, beta=beta)
