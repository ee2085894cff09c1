"""Thin tensor-level wrappers over the C-ABI (include/dyneval.h).

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic op on the product path is a
hand-written HIP kernel reached through ctypes.  Wrappers validate device/dtype/contiguity/alignment on the host so
a kernel never sees a shape it does not expect.  There is no CPU fallback: a CPU tensor raises DynError."""
import ctypes

import torch

from . import _lib
from ._lib import DynError, GemmDesc, check

_WS = {}
WORKSPACE_BYTES = 1 << 30
F32 = torch.float32
I32 = torch.int32


def _L():
    return _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype=F32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise DynError(f"{name}: expected a CUDA tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise DynError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t


def _cc(t, name, dtype=F32):
    _chk(t, name, dtype)
    if not t.is_contiguous():
        raise DynError(f"{name}: expected a contiguous tensor")
    if t.data_ptr() % 16:
        raise DynError(f"{name}: expected a 16-byte aligned tensor")
    return t


def _c(t, name, dtype=F32):
    """Contiguous but not necessarily 16-byte aligned (kernels with scalar accesses only)."""
    _chk(t, name, dtype)
    if not t.is_contiguous():
        raise DynError(f"{name}: expected a contiguous tensor")
    return t


def _opt(t, name, dtype=F32):
    return 0 if t is None else _cc(t, name, dtype).data_ptr()


class GroupParam:
    """A parameter (or its gradient) of a LOCKSTEP GROUP: R recordings advance through the same window step in one batch, each with its own
    adapted weights (model.SCConformerXL(group=R)).  `t` is the [R, *shape] view into the group's flat buffer, replicas `stride` elements
    apart.  Batches are ordered sample = chunk * R + replica, so sample s belongs to replica s % R."""
    __slots__ = ("t", "R", "stride")

    def __init__(self, t, R, stride):
        self.t, self.R, self.stride = t, int(R), int(stride)

    @property
    def shape(self):
        return self.t.shape[1:]

    def __getitem__(self, r):
        return self.t[r]

    def data_ptr(self):
        return self.t.data_ptr()


def _is_group(p):
    return isinstance(p, GroupParam)


_WS_OWNER = None   # scratch buffer of the model whose forward/backward is being launched or captured (use_workspace)


class use_workspace:
    """Route every op's scratch to `ws` while a model launches (or captures) its kernels.  A stream-keyed lookup is wrong
    inside a hipGraph capture: torch captures on its own side stream, so two models' graphs would be handed the same
    scratch buffer and race when their replays overlap on different streams."""

    def __init__(self, ws):
        self.ws = ws

    def __enter__(self):
        global _WS_OWNER
        self.prev, _WS_OWNER = _WS_OWNER, self.ws

    def __exit__(self, *exc):
        global _WS_OWNER
        _WS_OWNER = self.prev


def workspace(device=None):
    """Caller-owned scratch (split-K slabs, partial reductions, CTC lattice): the launching model's own buffer inside
    use_workspace(), otherwise one buffer per (device, stream) — concurrent recording chains must not share scratch."""
    if _WS_OWNER is not None:
        return _WS_OWNER
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
        _WS[key] = ws
        counters(ws)
    return ws


import os as _os

_COUNTERS = {}
N_COUNTERS = 16384
TICKET_DEFAULT = _os.environ.get("DYN_GEMM_TICKET", "0") == "1"


def counters(ws):
    """Arrival counters of the GEMM's in-kernel K-slice combine, one zeroed int32 buffer per scratch buffer (so per model replica /
    stream: concurrent chains never share them)."""
    key = ws.data_ptr()
    c = _COUNTERS.get(key)
    if c is None:
        c = torch.zeros(N_COUNTERS, dtype=torch.int32, device=ws.device)
        _COUNTERS[key] = c
    return c


# ----------------------------------------------------------------------------------------------- GEMM
def gemm(a, b, c, *, trans_a=False, trans_b=False, M, N, K, lda, ldb, ldc, alpha=1.0, beta=0.0, bias=None,
         nb1=1, nb2=1, sa=(0, 0), sb=(0, 0), sc=(0, 0), a_off=0, b_off=0, c_off=0, split_k=0, force=None, c_in=None,
         epilogue=0, aux=None, ticket=None, sbias=(0, 0)):
    """Raw strided batched GEMM (see dyn_gemm_desc). Offsets are in elements from each tensor's data_ptr.
    `force=(tile_m, tile_n, tail_slices)` pins the kernel configuration (autotuner / tests).
    `epilogue`: 0 none, EPI_SILU (C = silu(v), aux = v if given), EPI_SILU_GRAD (C = v * silu'(aux)); aux addressed like C."""
    _chk(a, "gemm.A"); _chk(b, "gemm.B"); _chk(c, "gemm.C")
    d = GemmDesc()
    if force is not None:
        d.tile_m, d.tile_n, d.tail_slices = force
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    d.M, d.N, d.K = M, N, K
    d.alpha, d.beta = alpha, beta
    d.A, d.lda, d.sa1, d.sa2 = a.data_ptr() + 4 * a_off, lda, sa[0], sa[1]
    d.B, d.ldb, d.sb1, d.sb2 = b.data_ptr() + 4 * b_off, ldb, sb[0], sb[1]
    d.C, d.ldc, d.sc1, d.sc2 = c.data_ptr() + 4 * c_off, ldc, sc[0], sc[1]
    d.bias = bias.data_ptr() if bias is not None else None
    d.bias_s1, d.bias_s2 = sbias
    d.C_in = (c_in.data_ptr() + 4 * c_off) if c_in is not None else None   # residual source (same addressing as C)
    d.nb1, d.nb2 = nb1, nb2
    d.split_k = split_k
    d.epilogue = int(epilogue)
    d.aux = (aux.data_ptr() + 4 * c_off) if aux is not None else None
    ws = workspace(c.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    if ticket is None:
        ticket = TICKET_DEFAULT
    if ticket:       # in-kernel combine of K slices instead of the separate reduce pass (opt-in: DYN_GEMM_TICKET=1; see gemm_f32.hip)
        cnt = counters(ws)
        d.counters, d.n_counters = cnt.data_ptr(), cnt.numel()
    prof = GEMM_PROFILE
    if prof is not None:
        prof["calls"] += 1
        prof["flops"] += 2.0 * M * N * K * nb1 * nb2
        prof["bytes"] += 4.0 * nb1 * nb2 * (M * K + K * N + M * N * (2 if beta != 0.0 else 1))  # algorithmic operand bytes
        if prof["calls"] % prof["every"] == 0:  # HIP events on the launch stream around this one launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(_L().dyn_gemm_f32(ctypes.byref(d), _stream()), "dyn_gemm_f32")
            e1.record()
            prof["samples_excl" if prof.get("exclusive_now") else "samples"].append((2.0 * M * N * K * nb1 * nb2, e0, e1))
            return c
    check(_L().dyn_gemm_f32(ctypes.byref(d), _stream()), "dyn_gemm_f32")
    return c


EPI_SILU, EPI_SILU_GRAD = 1, 2


def wgrad_desc(dy, x, dw, alpha=1.0, beta=1.0, colsum=None, colsum_beta=1.0):
    """Descriptor of dw[N, K] = alpha * dy[M, N]^T @ x[M, K] + beta * dw for gemm_grouped; `colsum` [N] also receives
    colsum_beta * colsum + column sums of dy (the bias gradient).  The tensors must stay alive and unmodified until the launch."""
    _cc(dy, "wgrad.dy"); _cc(x, "wgrad.x"); _cc(dw, "wgrad.dw")
    N, K = dw.shape
    M = dy.numel() // N
    d = GemmDesc()
    d.trans_a, d.trans_b = 1, 0
    d.M, d.N, d.K = N, K, M
    d.alpha, d.beta = alpha, beta
    d.A, d.lda = dy.data_ptr(), N
    d.B, d.ldb = x.data_ptr(), K
    d.C, d.ldc = dw.data_ptr(), K
    d.nb1 = d.nb2 = 1
    if colsum is not None:
        _cc(colsum, "wgrad.colsum")
        d.a_colsum, d.a_colsum_beta = colsum.data_ptr(), colsum_beta
    d._keep = (dy, x, dw, colsum)
    return d


MAX_GROUPS = 96     # kMaxGroups of csrc/gemm_f32.hip: descriptors per dyn_gemm_f32_grouped launch


def wgrad_groupable(dy, x, dw):
    """True when dw = dy^T @ x may ride in the grouped launch: dyn_gemm_f32_grouped has only the 16-byte vector staging, so it
    needs leading dimensions that are multiples of 4 (the single-GEMM path degrades to scalar staging instead of failing)."""
    N, K = dw.shape
    return N % 4 == 0 and K % 4 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0


def gemm_grouped(descs):
    """One launch over the 128x128 tiles of all `descs` (dyn_gemm_f32_grouped): the deferred weight gradients of a backward pass.
    More than MAX_GROUPS descriptors (8 per conformer block + 1: from 12 blocks on) go out in slices of MAX_GROUPS; the slices are
    stream-ordered, so they can share the descriptor-table workspace."""
    if len(descs) > MAX_GROUPS:
        for i in range(0, len(descs), MAX_GROUPS):
            gemm_grouped(descs[i:i + MAX_GROUPS])
        return
    n = len(descs)
    if n == 0:
        return
    arr = (GemmDesc * n)(*descs)
    ws = workspace()
    prof = GEMM_PROFILE
    fl = sum(2.0 * d.M * d.N * d.K for d in descs)
    if prof is not None:
        prof["calls"] += 1
        prof["flops"] += fl
        prof["bytes"] += sum(4.0 * (d.M * d.K + d.K * d.N + 2 * d.M * d.N) for d in descs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_L().dyn_gemm_f32_grouped(arr, n, ws.data_ptr(), ws.numel(), _stream()), "dyn_gemm_f32_grouped")
        e1.record()
        prof["samples_excl" if prof.get("exclusive_now") else "samples"].append((fl, e0, e1))
        return
    check(_L().dyn_gemm_f32_grouped(arr, n, ws.data_ptr(), ws.numel(), _stream()), "dyn_gemm_f32_grouped")


GEMM_PROFILE = None


def GEMM_PROFILE_EAGER():
    """True while bench.py's per-launch HIP-event sampling wants THIS model call to run eagerly (graph replays cannot be
    timed launch by launch): every `window_every`-th forward/backward pair of the timed region."""
    prof = GEMM_PROFILE
    if prof is None or not prof.get("window_every"):
        return False
    return prof["eager_now"]


def gemm_profile_start(every=8, window_every=0):
    """Sample every `every`-th GEMM launch with a HIP-event pair (bench.py's live roofline measurement).  With hipGraph
    replay enabled, `window_every` = n makes every n-th window step run eagerly (gemm_profile_tick) so its launches can
    be sampled; launch counters then cover the eager windows only."""
    global GEMM_PROFILE
    GEMM_PROFILE = {"calls": 0, "flops": 0.0, "bytes": 0.0, "every": int(every), "samples": [], "samples_excl": [],
                    "window_every": int(window_every), "windows": 0, "eager_now": False, "exclusive_now": False}


def gemm_profile_tick():
    """Called once per window step by the dynamic-eval loop: decides whether this step runs eagerly (sampled).
    -> 0 not sampled, 1 sampled as it runs (other chains may share the GPU), 2 sampled EXCLUSIVELY (the caller drains the
    device before and after, so each timed launch has the GPU to itself: the kernel's own duration).  Sampled steps
    alternate between the two kinds; with a single chain both are the same thing."""
    prof = GEMM_PROFILE
    if prof is None or not prof.get("window_every"):
        return 0
    kind = 0
    if prof["windows"] % prof["window_every"] == 0:
        kind = 2 if (prof["windows"] // prof["window_every"]) % 2 == 0 else 1
    prof["windows"] += 1
    return kind


def gemm_profile_mode(kind):
    """Set by the dynamic-eval loop right before each model call of a step (chains interleave on one host thread)."""
    prof = GEMM_PROFILE
    if prof is not None:
        prof["eager_now"] = kind > 0
        prof["exclusive_now"] = kind == 2


def gemm_profile_active():
    """True only while bench.py's sampling is on: the dynamic-eval loop enters its sampling blocks under this test and nowhere else."""
    return GEMM_PROFILE is not None


def gemm_profile_begin_step(device):
    """Start of a window step (or final-pass batch): -> 0 not sampled / 1 sampled shared / 2 sampled exclusively (device drained first)."""
    kind = gemm_profile_tick()
    if kind == 2:
        torch.cuda.synchronize(device)           # exclusive sample: the other chains' queued work drains first
    gemm_profile_mode(kind)
    return kind


def gemm_profile_before_yield(kind, ready_event):
    if kind == 2:
        ready_event.synchronize()                # keep the device to this chain until its forward has finished


def gemm_profile_resume_step(device, kind):
    """After the yield of a sampled step: other chains may have queued work meanwhile; the mode is per model call."""
    if kind == 2:
        torch.cuda.synchronize(device)
    gemm_profile_mode(kind)


def gemm_profile_end_step(device, kind):
    if kind == 2:
        torch.cuda.current_stream(device).synchronize()
    gemm_profile_mode(0)


def gemm_profile_stop():
    """-> dict(calls, flops, sampled, sampled_flops, sampled_ms): call after a stream synchronize."""
    global GEMM_PROFILE
    prof, GEMM_PROFILE = GEMM_PROFILE, None
    if prof is None:
        return None
    out = {"calls": prof["calls"], "flops": prof["flops"], "bytes": prof["bytes"], "attn_flops": prof.get("attn_flops", 0.0),
           "sampled_steps": (prof["windows"] + prof["window_every"] - 1) // prof["window_every"] if prof.get("window_every") else prof.get("windows", 0)}
    for key, name in (("samples", "shared"), ("samples_excl", "exclusive")):
        out[name] = {"sampled": len(prof[key]), "flops": sum(s[0] for s in prof[key]),
                     "ms": sum(s[1].elapsed_time(s[2]) for s in prof[key])}
    return out


def linear(x, w, bias=None, out=None, alpha=1.0, beta=0.0, residual=None, epilogue=0, aux=None):
    """out[M, N] = alpha * x[M, K] @ w[N, K]^T + beta * (residual if given else out) + bias   (torch.nn.Linear layout);
    with epilogue=EPI_SILU: out = silu(that) and aux (if given) = that."""
    if _is_group(w):
        return _linear_group(x, w, bias, out, alpha, beta, residual, epilogue, aux)
    _cc(x, "linear.x"); _cc(w, "linear.w")
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    assert w.shape[1] == K, (w.shape, K)
    if out is None:
        assert beta == 0.0 or residual is not None
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=F32)
    return gemm(x, w, out, trans_b=True, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias, alpha=alpha, beta=beta, c_in=residual,
                epilogue=epilogue, aux=aux)


def _group_dims(x, R, what):
    """x [S, ..., K] with S = chunks * R samples (sample = chunk * R + replica) -> (chunks, rows per sample)."""
    S = x.shape[0]
    if S % R:
        raise DynError(f"{what}: batch {S} is not a multiple of the group size {R}")
    return S // R, x.numel() // (S * x.shape[-1])


def _linear_group(x, w, bias, out, alpha, beta, residual, epilogue, aux):
    """linear() over a lockstep group: ONE batched product, batch = (chunk, replica); replica r multiplies its own rows with its own weight
    (B operand and bias step through the group's flat parameter buffer with the replica stride)."""
    _cc(x, "linear.x")
    R = w.R
    N, K = w.shape
    assert x.shape[-1] == K, (x.shape, K)
    nch, M = _group_dims(x, R, "linear")
    if out is None:
        assert beta == 0.0 or residual is not None
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=F32)
    return gemm(x, w.t, out, trans_b=True, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=None if bias is None else bias.t, alpha=alpha, beta=beta,
                c_in=residual, epilogue=epilogue, aux=aux, nb1=nch, nb2=R, sa=(R * M * K, M * K), sb=(0, w.stride), sc=(R * M * N, M * N),
                sbias=(0, 0 if bias is None else bias.stride))


def linear_dgrad(dy, w, out=None, alpha=1.0, beta=0.0, epilogue=0, aux=None):
    """dx[M, K] = alpha * dy[M, N] @ w[N, K] + beta * dx; with epilogue=EPI_SILU_GRAD: dx *= silu'(aux) (aux = the pre-activation)."""
    if _is_group(w):
        _cc(dy, "linear_dgrad.dy")
        R = w.R
        N, K = w.shape
        nch, M = _group_dims(dy, R, "linear_dgrad")
        if out is None:
            assert beta == 0.0
            out = torch.empty(*dy.shape[:-1], K, device=dy.device, dtype=F32)
        return gemm(dy, w.t, out, M=M, N=K, K=N, lda=N, ldb=K, ldc=K, alpha=alpha, beta=beta, epilogue=epilogue, aux=aux, nb1=nch, nb2=R,
                    sa=(R * M * N, M * N), sb=(0, w.stride), sc=(R * M * K, M * K))
    _cc(dy, "linear_dgrad.dy"); _cc(w, "linear_dgrad.w")
    N, K = w.shape
    M = dy.numel() // N
    if out is None:
        assert beta == 0.0
        out = torch.empty(*dy.shape[:-1], K, device=dy.device, dtype=F32)
    return gemm(dy, w, out, M=M, N=K, K=N, lda=N, ldb=K, ldc=K, alpha=alpha, beta=beta, epilogue=epilogue, aux=aux)


def linear_wgrad(dy, x, dw, alpha=1.0, beta=1.0):
    """dw[N, K] = alpha * dy[M, N]^T @ x[M, K] + beta * dw   (deterministic split-K)."""
    _cc(dy, "linear_wgrad.dy"); _cc(x, "linear_wgrad.x"); _cc(dw, "linear_wgrad.dw")
    N, K = dw.shape
    M = dy.numel() // N
    return gemm(dy, x, dw, trans_a=True, M=N, N=K, K=M, lda=N, ldb=K, ldc=K, alpha=alpha, beta=beta)


# ----------------------------------------------------------------------------------------------- elementwise
def silu(x, out=None):
    _cc(x, "silu.x")
    out = torch.empty_like(x) if out is None else _cc(out, "silu.out")
    check(_L().dyn_silu_fwd(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "dyn_silu_fwd")
    return out


def silu_bwd(x, dy, out=None):
    _cc(x, "silu_bwd.x"); _cc(dy, "silu_bwd.dy")
    out = torch.empty_like(x) if out is None else _cc(out, "silu_bwd.out")
    check(_L().dyn_silu_bwd(x.data_ptr(), dy.data_ptr(), out.data_ptr(), x.numel(), _stream()), "dyn_silu_bwd")
    return out


def glu(u, out=None):
    _cc(u, "glu.u")
    C = u.shape[-1] // 2
    rows = u.numel() // (2 * C)
    out = torch.empty(*u.shape[:-1], C, device=u.device, dtype=F32) if out is None else _cc(out, "glu.out")
    check(_L().dyn_glu_fwd(u.data_ptr(), out.data_ptr(), rows, C, _stream()), "dyn_glu_fwd")
    return out


def glu_bwd(u, dy, out=None):
    _cc(u, "glu_bwd.u"); _cc(dy, "glu_bwd.dy")
    C = u.shape[-1] // 2
    rows = u.numel() // (2 * C)
    out = torch.empty_like(u) if out is None else _cc(out, "glu_bwd.out")
    check(_L().dyn_glu_bwd(u.data_ptr(), dy.data_ptr(), out.data_ptr(), rows, C, _stream()), "dyn_glu_bwd")
    return out


def axpby(x, y, a=1.0, b=1.0):
    """y = a * x + b * y (in place on y)."""
    _cc(x, "axpby.x"); _cc(y, "axpby.y")
    assert x.numel() == y.numel()
    check(_L().dyn_axpby(x.data_ptr(), y.data_ptr(), a, b, x.numel(), _stream()), "dyn_axpby")
    return y


def reduce_partials(partial, out, beta=1.0):
    """out[n] = beta * out[n] + sum_p partial[p, n] (deterministic: p in increasing order per column)."""
    _cc(partial, "reduce_partials.partial"); _cc(out, "reduce_partials.out")
    n = out.numel()
    P = partial.numel() // n
    assert P * n == partial.numel()
    check(_L().dyn_reduce_partials(partial.data_ptr(), out.data_ptr(), P, n, beta, _stream()), "dyn_reduce_partials")
    return out


DEFER_ARENA_BYTES = 160 << 20


class reduce_defer:
    """`with ops.reduce_defer(arena):` — the weight-gradient / bias-sum reductions of the backward kernels launched inside are recorded
    and run as one launch per 96 at exit (dyn_reduce_defer_begin / _flush; bit-identical).  Nothing inside may read those gradients; the
    block must not span a `yield` (the context is per host thread).  `arena=None`: no deferral."""

    def __init__(self, arena):
        self.arena = arena

    def __enter__(self):
        if self.arena is not None:
            check(_L().dyn_reduce_defer_begin(self.arena.data_ptr(), self.arena.numel()), "dyn_reduce_defer_begin")
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.arena is None:
            return False
        if exc_type is not None:
            _L().dyn_reduce_defer_abort()
            return False
        check(_L().dyn_reduce_defer_flush(_stream()), "dyn_reduce_defer_flush")
        return False


def colsum(x, out, beta=1.0):
    """out[C] = beta * out + sum over rows of x[rows, C]."""
    _cc(x, "colsum.x"); _cc(out, "colsum.out")
    C = out.numel()
    rows = x.numel() // C
    ws = workspace(x.device)
    check(_L().dyn_colsum(x.data_ptr(), out.data_ptr(), rows, C, beta, ws.data_ptr(), ws.numel(), _stream()), "dyn_colsum")
    return out


def transpose_ft(x, out=None):
    """[F, T] window view (T contiguous, arbitrary row stride) -> contiguous [T, F]."""
    _chk(x, "transpose_ft.x")
    F, T = x.shape
    assert x.stride(1) == 1
    out = torch.empty(T, F, device=x.device, dtype=F32) if out is None else _cc(out, "transpose_ft.out")
    check(_L().dyn_transpose_ft(x.data_ptr(), out.data_ptr(), F, T, x.stride(0), _stream()), "dyn_transpose_ft")
    return out


def specaug_freqmask(x, f0, width, value=0.0):
    """In-place frequency masks on a contiguous [F, T] window; f0/width are int32 CUDA tensors; `value` may be a float or
    a 1-element CUDA tensor (device-side fill value: no host round trip)."""
    _cc(x, "specaug.x"); _cc(f0, "specaug.f0", I32); _cc(width, "specaug.width", I32)
    F, T = x.shape
    vdev = value.data_ptr() if isinstance(value, torch.Tensor) else 0
    check(_L().dyn_specaug_freqmask(x.data_ptr(), F, T, f0.data_ptr(), width.data_ptr(), f0.numel(), 0.0 if vdev else value, vdev, _stream()),
          "dyn_specaug_freqmask")
    return x


def specaug_timemask(x, t0, width, value=0.0):
    """In-place time masks on a contiguous [F, T] window."""
    _cc(x, "specaug.x"); _cc(t0, "specaug.t0", I32); _cc(width, "specaug.width", I32)
    F, T = x.shape
    vdev = value.data_ptr() if isinstance(value, torch.Tensor) else 0
    check(_L().dyn_specaug_timemask(x.data_ptr(), F, T, t0.data_ptr(), width.data_ptr(), t0.numel(), 0.0 if vdev else value, vdev, _stream()),
          "dyn_specaug_timemask")
    return x


ATTN_NSPLIT_DEFAULT = int(_os.environ.get("DYN_ATTN_NSPLIT", "0"))


def attention_fwd(qkv, B, T, H, D, scale, out=None, want_lse=False, nsplit=0):
    """Fused attention on the packed [B, T, 3 * H * D] QKV activation (q | k | v, head h at +h * D) -> [B, T, H * D];
    with `want_lse` (grad mode) returns (out, lse [B, H, T]) for attention_bwd.  `nsplit`: key splits per query block (0 = chosen
    by the library from the launch size, 1 = none): launches that do not fill the chip get nsplit x the workgroups, the partial
    outputs go through the caller's workspace and are merged in split order."""
    _cc(qkv, "attention.qkv")
    HD = H * D
    out = torch.empty(B, T, HD, device=qkv.device, dtype=F32) if out is None else out
    if GEMM_PROFILE is not None:
        GEMM_PROFILE["attn_flops"] = GEMM_PROFILE.get("attn_flops", 0.0) + 4.0 * B * H * T * T * D   # matrix-core work outside dyn_gemm_f32
    base = qkv.data_ptr()
    lse = torch.empty(B, H, T, device=qkv.device, dtype=F32) if want_lse else None
    if nsplit == 0 and ATTN_NSPLIT_DEFAULT:
        nsplit = ATTN_NSPLIT_DEFAULT     # A/B switch only (DYN_ATTN_NSPLIT=1: never split)
    ws = workspace(qkv.device)
    need = _L().dyn_attention_fwd_split_workspace_bytes(B, T, H, nsplit)
    if need > ws.numel():
        nsplit = 1                      # scratch too small for the slabs (very long windows): the unsplit launch needs none
    check(_L().dyn_attention_fwd_split(base, base + 4 * HD, base + 8 * HD, out.data_ptr(), 0 if lse is None else lse.data_ptr(), B, T, H, D,
                                       3 * HD, T * 3 * HD, HD, T * HD, scale, nsplit, ws.data_ptr(), ws.numel(), _stream()), "dyn_attention_fwd_split")
    return (out, lse) if want_lse else out


def attention_bwd(qkv, out, dout, lse, B, T, H, D, scale, dqkv=None):
    """Backward of attention_fwd: (packed qkv, out, dout [B, T, H * D], lse [B, H, T]) -> packed dqkv [B, T, 3 * H * D]."""
    _cc(qkv, "attention_bwd.qkv"); _cc(out, "attention_bwd.out"); _cc(dout, "attention_bwd.dout"); _cc(lse, "attention_bwd.lse")
    HD = H * D
    dqkv = torch.empty_like(qkv) if dqkv is None else _cc(dqkv, "attention_bwd.dqkv")
    delta = torch.empty(B, H, T, device=qkv.device, dtype=F32)
    if GEMM_PROFILE is not None:
        GEMM_PROFILE["attn_flops"] = GEMM_PROFILE.get("attn_flops", 0.0) + 14.0 * B * H * T * T * D   # 3 + 4 products (P is re-formed twice)
    base, g = qkv.data_ptr(), dqkv.data_ptr()
    check(_L().dyn_attention_bwd(base, base + 4 * HD, base + 8 * HD, out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                 g, g + 4 * HD, g + 8 * HD, B, T, H, D, 3 * HD, T * 3 * HD, HD, T * HD, 3 * HD, T * 3 * HD, scale, _stream()),
          "dyn_attention_bwd")
    return dqkv


# ----------------------------------------------------------------------------------------------- norms
def layernorm(x, gamma, beta, eps=1e-5, out=None):
    _cc(x, "layernorm.x")
    C = x.shape[-1]
    rows = x.numel() // C
    out = torch.empty_like(x) if out is None else _cc(out, "layernorm.out")
    mean = torch.empty(rows, device=x.device, dtype=F32)
    rstd = torch.empty(rows, device=x.device, dtype=F32)
    if _is_group(gamma):       # lockstep group: sample s of x [S, T, C] is normalised with replica s % R's weights, one launch
        _, rps = _group_dims(x, gamma.R, "layernorm")
        check(_L().dyn_layernorm_fwd_g(x.data_ptr(), gamma.data_ptr(), 0 if beta is None else beta.data_ptr(), out.data_ptr(), mean.data_ptr(),
                                       rstd.data_ptr(), rows, C, eps, rps, gamma.R, gamma.stride, _stream()), "dyn_layernorm_fwd_g")
        return out, mean, rstd
    _cc(gamma, "layernorm.gamma")
    check(_L().dyn_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), _opt(beta, "layernorm.beta"), out.data_ptr(),
                                 mean.data_ptr(), rstd.data_ptr(), rows, C, eps, _stream()), "dyn_layernorm_fwd")
    return out, mean, rstd


def layernorm_bwd(x, gamma, mean, rstd, dy, dx, dgamma, dbeta, dx_beta=0.0, wgrad_beta=1.0, dx_in=None):
    """dx = norm_bwd(dy) + dx_beta * (dx_in if given else dx)."""
    _cc(x, "layernorm_bwd.x"); _cc(dy, "layernorm_bwd.dy"); _cc(dx, "layernorm_bwd.dx")
    C = x.shape[-1]
    rows = x.numel() // C
    ws = workspace(x.device)
    if _is_group(gamma):
        _, rps = _group_dims(x, gamma.R, "layernorm_bwd")
        src = dx if dx_in is None else _cc(dx_in, "layernorm_bwd.dx_in")
        check(_L().dyn_layernorm_bwd_g(x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(), src.data_ptr(),
                                       dx.data_ptr(), dx_beta, 0 if dgamma is None else dgamma.data_ptr(), 0 if dbeta is None else dbeta.data_ptr(),
                                       wgrad_beta, rows, C, rps, gamma.R, gamma.stride, ws.data_ptr(), ws.numel(), _stream()), "dyn_layernorm_bwd_g")
        return dx
    if dx_in is not None:
        _cc(dx_in, "layernorm_bwd.dx_in")
        check(_L().dyn_layernorm_bwd_res(x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dx_in.data_ptr(),
                                         dx.data_ptr(), dx_beta, _opt(dgamma, "dgamma"), _opt(dbeta, "dbeta"), wgrad_beta, rows, C,
                                         ws.data_ptr(), ws.numel(), _stream()), "dyn_layernorm_bwd_res")
        return dx
    check(_L().dyn_layernorm_bwd(x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(),
                                 dx.data_ptr(), dx_beta, _opt(dgamma, "dgamma"), _opt(dbeta, "dbeta"), wgrad_beta, rows, C,
                                 ws.data_ptr(), ws.numel(), _stream()), "dyn_layernorm_bwd")
    return dx


def chanaffine(x, mean, var, weight, bias, eps=1e-5, out=None):
    """BatchRenorm1d in eval mode: (x - mean_c) * rsqrt(var_c + eps) * weight_c + bias_c over the last dim."""
    _c(x, "chanaffine.x")
    C = x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    check(_L().dyn_chanaffine_fwd(x.data_ptr(), mean.data_ptr(), var.data_ptr(), weight.data_ptr(), _opt(bias, "bias") if bias is not None else 0,
                                  out.data_ptr(), x.numel() // C, C, eps, _stream()), "dyn_chanaffine_fwd")
    return out


def chanaffine_bwd(x, mean, var, weight, dy, dx, dweight, dbias, eps=1e-5, dx_beta=0.0, wgrad_beta=1.0):
    _c(x, "chanaffine_bwd.x"); _c(dy, "chanaffine_bwd.dy"); _c(dx, "chanaffine_bwd.dx")
    C = x.shape[-1]
    ws = workspace(x.device)
    check(_L().dyn_chanaffine_bwd(x.data_ptr(), mean.data_ptr(), var.data_ptr(), weight.data_ptr(), dy.data_ptr(), dx.data_ptr(), dx_beta,
                                  0 if dweight is None else dweight.data_ptr(), 0 if dbias is None else dbias.data_ptr(), wgrad_beta,
                                  x.numel() // C, C, eps, ws.data_ptr(), ws.numel(), _stream()), "dyn_chanaffine_bwd")
    return dx


def rmsnorm(x, gamma, eps=1e-5, out=None):
    _cc(x, "rmsnorm.x"); _cc(gamma, "rmsnorm.gamma")
    C = x.shape[-1]
    rows = x.numel() // C
    out = torch.empty_like(x) if out is None else _cc(out, "rmsnorm.out")
    rstd = torch.empty(rows, device=x.device, dtype=F32)
    check(_L().dyn_rmsnorm_fwd(x.data_ptr(), gamma.data_ptr(), out.data_ptr(), rstd.data_ptr(), rows, C, eps, _stream()),
          "dyn_rmsnorm_fwd")
    return out, rstd


def rmsnorm_bwd(x, gamma, rstd, dy, dx, dgamma, dx_beta=0.0, wgrad_beta=1.0):
    _cc(x, "rmsnorm_bwd.x"); _cc(dy, "rmsnorm_bwd.dy"); _cc(dx, "rmsnorm_bwd.dx")
    C = x.shape[-1]
    rows = x.numel() // C
    ws = workspace(x.device)
    if _is_group(gamma):
        _, rps = _group_dims(x, gamma.R, "rmsnorm_bwd")
        check(_L().dyn_rmsnorm_bwd_g(x.data_ptr(), gamma.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dx.data_ptr(), dx_beta,
                                     0 if dgamma is None else dgamma.data_ptr(), wgrad_beta, rows, C, rps, gamma.R, gamma.stride,
                                     ws.data_ptr(), ws.numel(), _stream()), "dyn_rmsnorm_bwd_g")
        return dx
    check(_L().dyn_rmsnorm_bwd(x.data_ptr(), gamma.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dx.data_ptr(), dx_beta,
                               _opt(dgamma, "dgamma"), wgrad_beta, rows, C, ws.data_ptr(), ws.numel(), _stream()),
          "dyn_rmsnorm_bwd")
    return dx


# ----------------------------------------------------------------------------------------------- softmax
def _rows_L(x):
    L = x.shape[-1]
    return x.numel() // L, L


def softmax(x, out=None, valid=None):
    """Row softmax; `valid` (device int32 scalar): columns >= valid are masked keys (probability 0)."""
    _cc(x, "softmax.x")
    rows, L = _rows_L(x)
    out = torch.empty_like(x) if out is None else _cc(out, "softmax.out")
    if valid is not None:
        check(_L().dyn_softmax_fwd_len(x.data_ptr(), out.data_ptr(), rows, L, L, L, _valid_ptr(valid, "softmax"), _stream()), "dyn_softmax_fwd_len")
        return out
    check(_L().dyn_softmax_fwd(x.data_ptr(), out.data_ptr(), rows, L, L, L, _stream()), "dyn_softmax_fwd")
    return out


def softmax_bwd(y, dy, out=None, scale=1.0):
    _cc(y, "softmax_bwd.y"); _cc(dy, "softmax_bwd.dy")
    rows, L = _rows_L(y)
    out = torch.empty_like(y) if out is None else _cc(out, "softmax_bwd.out")
    check(_L().dyn_softmax_bwd(y.data_ptr(), dy.data_ptr(), out.data_ptr(), rows, L, L, scale, _stream()), "dyn_softmax_bwd")
    return out


def log_softmax(x, out=None):
    _cc(x, "log_softmax.x")
    rows, L = _rows_L(x)
    out = torch.empty_like(x) if out is None else _cc(out, "log_softmax.out")
    check(_L().dyn_log_softmax_fwd(x.data_ptr(), out.data_ptr(), rows, L, L, L, _stream()), "dyn_log_softmax_fwd")
    return out


def log_softmax_bwd(y, dy, out=None):
    _cc(y, "log_softmax_bwd.y"); _cc(dy, "log_softmax_bwd.dy")
    rows, L = _rows_L(y)
    out = torch.empty_like(y) if out is None else _cc(out, "log_softmax_bwd.out")
    check(_L().dyn_log_softmax_bwd(y.data_ptr(), dy.data_ptr(), out.data_ptr(), rows, L, L, _stream()), "dyn_log_softmax_bwd")
    return out


def entropy_grad(log_probs, scale):
    """Gradient of `scale * sum_rows H(row)` w.r.t. log-probs [..., C]; returns (grad, per-row entropies)."""
    _cc(log_probs, "entropy_grad.log_probs")
    rows, L = _rows_L(log_probs)
    g = torch.empty_like(log_probs)
    ent = torch.empty(rows, device=log_probs.device, dtype=F32)
    check(_L().dyn_entropy_grad(log_probs.data_ptr(), g.data_ptr(), ent.data_ptr(), rows, L, L, scale, _stream()), "dyn_entropy_grad")
    return g, ent


def conv2d_first_dgrad(dz, w, T, F):
    """dz [B, To, Fo, C] -> dx [B, T, F] for the 1-channel 3x3/s2 first conv."""
    _cc(dz, "conv2d_first_dgrad.dz"); _cc(w, "conv2d_first_dgrad.w")
    B, C = dz.shape[0], dz.shape[-1]
    dx = torch.empty(B, T, F, device=dz.device, dtype=F32)
    check(_L().dyn_conv2d_first_dgrad(dz.data_ptr(), w.data_ptr(), dx.data_ptr(), B, T, F, C, _stream()), "dyn_conv2d_first_dgrad")
    return dx


# ----------------------------------------------------------------------------------------------- convolutions
def dwconv1d(x, w, bias, out=None):
    """x [B, T, C] channels-last, w [C, KW]."""
    _cc(x, "dwconv1d.x"); _cc(w, "dwconv1d.w")
    B, T, C = x.shape
    out = torch.empty_like(x) if out is None else _cc(out, "dwconv1d.out")
    check(_L().dyn_dwconv1d_fwd(x.data_ptr(), w.data_ptr(), _opt(bias, "bias"), out.data_ptr(), B, T, C, w.shape[1], _stream()),
          "dyn_dwconv1d_fwd")
    return out


def dwconv1d_dgrad(dy, w, out=None, beta=0.0):
    _cc(dy, "dwconv1d_dgrad.dy")
    B, T, C = dy.shape
    out = torch.empty_like(dy) if out is None else _cc(out, "dwconv1d_dgrad.out")
    if _is_group(w):
        _group_dims(dy, w.R, "dwconv1d_dgrad")
        check(_L().dyn_dwconv1d_dgrad_g(dy.data_ptr(), w.data_ptr(), out.data_ptr(), B, T, C, w.shape[1], beta, w.R, w.stride, _stream()),
              "dyn_dwconv1d_dgrad_g")
        return out
    _cc(w, "dwconv1d_dgrad.w")
    check(_L().dyn_dwconv1d_dgrad(dy.data_ptr(), w.data_ptr(), out.data_ptr(), B, T, C, w.shape[1], beta, _stream()),
          "dyn_dwconv1d_dgrad")
    return out


def dwconv1d_wgrad(x, dy, dw, dbias, beta=1.0):
    _cc(x, "dwconv1d_wgrad.x"); _cc(dy, "dwconv1d_wgrad.dy")
    B, T, C = x.shape
    ws = workspace(x.device)
    if _is_group(dw):
        _group_dims(x, dw.R, "dwconv1d_wgrad")
        check(_L().dyn_dwconv1d_wgrad_g(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0 if dbias is None else dbias.data_ptr(), beta, B, T, C,
                                        dw.shape[1], dw.R, dw.stride, ws.data_ptr(), ws.numel(), _stream()), "dyn_dwconv1d_wgrad_g")
        return
    _cc(dw, "dwconv1d_wgrad.dw")
    check(_L().dyn_dwconv1d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _opt(dbias, "dbias"), beta, B, T, C,
                                  dw.shape[1], ws.data_ptr(), ws.numel(), _stream()), "dyn_dwconv1d_wgrad")


def convmod_fwd(u, w, bias, gamma, beta, layernorm, eps, save):
    """Fused GLU -> dwconv(k=9) -> norm -> SiLU.  u [B, T, 2C] -> s [B, T, C]; with `save` also (g, c, nn, mean, rstd).
    Lockstep group (w a GroupParam): one launch per sample into shared output tensors (sample s: replica s % R's weights)."""
    _cc(u, "convmod.u")
    B, T, C2 = u.shape
    C = C2 // 2
    dev = u.device
    s = torch.empty(B, T, C, device=dev, dtype=F32)
    g = c = nn = mean = rstd = None
    if save:
        g, c, nn = (torch.empty(B, T, C, device=dev, dtype=F32) for _ in range(3))
        rstd = torch.empty(B * T, device=dev, dtype=F32)
        mean = torch.empty(B * T, device=dev, dtype=F32) if layernorm else None
    if _is_group(w):        # lockstep group: one launch, sample k takes replica k % R's filters and norm weights
        _group_dims(u, w.R, "convmod")
        check(_L().dyn_convmod_fwd_g(u.data_ptr(), w.data_ptr(), 0 if bias is None else bias.data_ptr(), gamma.data_ptr(),
                                     0 if beta is None else beta.data_ptr(), s.data_ptr(), _opt(g, "g"), _opt(c, "c"), _opt(nn, "nn"),
                                     _opt(mean, "mean"), _opt(rstd, "rstd"), B, T, C, w.shape[1], int(layernorm), eps, w.R, w.stride, _stream()),
              "dyn_convmod_fwd_g")
        return s, g, c, nn, mean, rstd
    _cc(w, "convmod.w")
    check(_L().dyn_convmod_fwd(u.data_ptr(), w.data_ptr(), _opt(bias, "bias"), gamma.data_ptr(), _opt(beta, "beta"), s.data_ptr(),
                               _opt(g, "g"), _opt(c, "c"), _opt(nn, "nn"), _opt(mean, "mean"), _opt(rstd, "rstd"), B, T, C, w.shape[1],
                               int(layernorm), eps, _stream()), "dyn_convmod_fwd")
    return s, g, c, nn, mean, rstd


def out_len(n):
    """Output length of a 3-wide, stride-2, pad-1 convolution."""
    return (n - 1) // 2 + 1


def conv2d_first(x, w, bias, out=None):
    """x [B, T, F] -> z [B, To, Fo, C]; w [C, 3, 3]."""
    _cc(x, "conv2d_first.x"); _cc(w, "conv2d_first.w"); _cc(bias, "conv2d_first.bias")
    B, T, F = x.shape
    C = w.shape[0]
    if out is None:
        out = torch.empty(B, out_len(T), out_len(F), C, device=x.device, dtype=F32)
    check(_L().dyn_conv2d_first_fwd(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), B, T, F, C, _stream()),
          "dyn_conv2d_first_fwd")
    return out


def conv2d_first_wgrad(x, dz, dw, dbias, beta=1.0):
    _cc(x, "conv2d_first_wgrad.x"); _cc(dz, "conv2d_first_wgrad.dz")
    B, T, F = x.shape
    C = dw.shape[0]
    ws = workspace(x.device)
    check(_L().dyn_conv2d_first_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), dbias.data_ptr(), beta, B, T, F, C,
                                      ws.data_ptr(), ws.numel(), _stream()), "dyn_conv2d_first_wgrad")


def dwconv2d_s2(z, w, bias, out=None):
    """u = bias + dw3x3_s2(silu(z)); z [B, T, F, C] -> [B, To, Fo, C]; w [C, 3, 3]."""
    _cc(z, "dwconv2d_s2.z"); _cc(w, "dwconv2d_s2.w"); _cc(bias, "dwconv2d_s2.bias")
    B, T, F, C = z.shape
    if out is None:
        out = torch.empty(B, out_len(T), out_len(F), C, device=z.device, dtype=F32)
    check(_L().dyn_dwconv2d_s2_fwd(z.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), B, T, F, C, _stream()),
          "dyn_dwconv2d_s2_fwd")
    return out


def dwconv2d_s2_dgrad(z, w, du, out=None):
    _cc(z, "dwconv2d_s2_dgrad.z"); _cc(du, "dwconv2d_s2_dgrad.du")
    B, T, F, C = z.shape
    out = torch.empty_like(z) if out is None else _cc(out, "dwconv2d_s2_dgrad.out")
    check(_L().dyn_dwconv2d_s2_dgrad(z.data_ptr(), w.data_ptr(), du.data_ptr(), out.data_ptr(), B, T, F, C, _stream()),
          "dyn_dwconv2d_s2_dgrad")
    return out


def dwconv2d_s2_wgrad(z, du, dw, dbias, beta=1.0):
    _cc(z, "dwconv2d_s2_wgrad.z"); _cc(du, "dwconv2d_s2_wgrad.du")
    B, T, F, C = z.shape
    ws = workspace(z.device)
    check(_L().dyn_dwconv2d_s2_wgrad(z.data_ptr(), du.data_ptr(), dw.data_ptr(), dbias.data_ptr(), beta, B, T, F, C,
                                     ws.data_ptr(), ws.numel(), _stream()), "dyn_dwconv2d_s2_wgrad")


def sub12_fwd(x, w1, b1, w2, b2, out=None):
    """Fused first two subsampling stages: x [B, T, F] -> u2 [B, T2, F2, C] = b2 + dw3x3_s2(silu(b1 + conv3x3_s2(x))); the
    [B, T1, F1, C] intermediate never reaches HBM (dyn_sub12_fwd)."""
    _cc(x, "sub12.x"); _cc(w1, "sub12.w1"); _cc(b1, "sub12.b1"); _cc(w2, "sub12.w2"); _cc(b2, "sub12.b2")
    B, T, F = x.shape
    C = w1.shape[0]
    if out is None:
        out = torch.empty(B, out_len(out_len(T)), out_len(out_len(F)), C, device=x.device, dtype=F32)
    check(_L().dyn_sub12_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), out.data_ptr(), B, T, F, C, _stream()),
          "dyn_sub12_fwd")
    return out


def sub12_bwd(x, du2, w1, b1, w2, dw1, db1, dw2, db2, beta=1.0):
    """Weight / bias gradients of both fused stages from du2 [B, T2, F2, C] (z1 recomputed from x, dz1 never stored)."""
    _cc(x, "sub12_bwd.x"); _cc(du2, "sub12_bwd.du2")
    B, T, F = x.shape
    C = w1.shape[0]
    ws = workspace(x.device)
    check(_L().dyn_sub12_bwd(x.data_ptr(), du2.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), dw1.data_ptr(), db1.data_ptr(),
                             dw2.data_ptr(), db2.data_ptr(), beta, B, T, F, C, ws.data_ptr(), ws.numel(), _stream()), "dyn_sub12_bwd")


def rotary(x, cos, sin, B, T, n_heads, D, row_stride, inverse=False):
    """In place on the first n_heads*D floats of every row_stride-float row of x."""
    _cc(x, "rotary.x"); _cc(cos, "rotary.cos"); _cc(sin, "rotary.sin")
    assert cos.shape[0] >= T and cos.shape[1] == D // 2
    check(_L().dyn_rotary(x.data_ptr(), cos.data_ptr(), sin.data_ptr(), B, T, n_heads, D, row_stride, int(inverse), _stream()),
          "dyn_rotary")
    return x


# ----------------------------------------------------------------------------------------------- CTC
def ctc_greedy(log_probs, blank):
    """log_probs [B, T, C] (or [T, C]) on device -> (ids int32 [B, T] device, lengths int32 [B] device).
    Only out_len[b] leading ids of each row are valid."""
    _c(log_probs, "ctc_greedy.log_probs")
    lp = log_probs if log_probs.dim() == 3 else log_probs.unsqueeze(0)
    B, T, C = lp.shape
    arg = torch.empty(B, T, device=lp.device, dtype=I32)
    ids = torch.empty(B, T, device=lp.device, dtype=I32)
    n = torch.empty(B, device=lp.device, dtype=I32)
    check(_L().dyn_ctc_greedy(lp.data_ptr(), B, T, C, C, blank, arg.data_ptr(), ids.data_ptr(), n.data_ptr(), _stream()),
          "dyn_ctc_greedy")
    return ids, n


def argmax_rows(x):
    """x [rows, C] -> (ids int32 [rows], max values [rows]); first maximum wins."""
    _c(x, "argmax_rows.x")
    rows, C = x.shape
    ids = torch.empty(rows, device=x.device, dtype=I32)
    vals = torch.empty(rows, device=x.device, dtype=F32)
    check(_L().dyn_argmax_rows(x.data_ptr(), rows, C, C, ids.data_ptr(), vals.data_ptr(), _stream()), "dyn_argmax_rows")
    return ids, vals


def ctc_loss(log_probs, targets, input_lengths, target_lengths, blank, reduction="sum", grad_scale=1.0, want_grad=True):
    """log_probs [B, T, C] contiguous; targets int32 [B, S_max]; lengths int32 [B] (all on device).
    Returns (loss [1] device tensor, nll [B], grad [B, T, C] or None) with torch.nn.CTCLoss semantics."""
    _c(log_probs, "ctc_loss.log_probs"); _c(targets, "ctc_loss.targets", I32)
    _cc(input_lengths, "ctc_loss.input_lengths", I32); _cc(target_lengths, "ctc_loss.target_lengths", I32)
    B, T, C = log_probs.shape
    S_max = targets.shape[1] if targets.numel() else 0
    loss = torch.empty(1, device=log_probs.device, dtype=F32)
    nll = torch.empty(B, device=log_probs.device, dtype=F32)
    grad = torch.empty_like(log_probs) if want_grad else None
    ws = workspace(log_probs.device)
    need = _L().dyn_ctc_loss_workspace_bytes(T, B, S_max)
    if need > ws.numel():
        raise DynError(f"ctc_loss: workspace {ws.numel()} < {need} bytes")
    tptr = targets.data_ptr() if targets.numel() else nll.data_ptr()
    check(_L().dyn_ctc_loss(log_probs.data_ptr(), T, B, C, C, T * C, tptr, S_max, input_lengths.data_ptr(),
                            target_lengths.data_ptr(), blank, {"sum": 0, "mean": 1}[reduction], grad_scale, loss.data_ptr(),
                            nll.data_ptr(), 0 if grad is None else grad.data_ptr(), C, T * C, ws.data_ptr(), ws.numel(),
                            _stream()), "dyn_ctc_loss")
    return loss, nll, grad


def ctc_lattice(T, B, S_max, device):
    """Views of the lattice the last ops.ctc_loss call of this (device, stream) left in the workspace:
    (alpha [B, T, L], beta [B, T, L], nll [B]) with L = 2 * max(S_max, 1) + 1; beta is only written when a gradient was asked for."""
    import ctypes
    off = (ctypes.c_int64 * 4)()
    L = ctypes.c_int64()
    check(_L().dyn_ctc_loss_workspace_layout(T, B, S_max, ctypes.cast(off, ctypes.c_void_p), ctypes.cast(ctypes.byref(L), ctypes.c_void_p)),
          "dyn_ctc_loss_workspace_layout")
    ws = workspace(device)
    n = B * T * L.value * 4
    alpha = ws[off[1]:off[1] + n].view(F32).view(B, T, L.value)
    beta = ws[off[2]:off[2] + n].view(F32).view(B, T, L.value)
    nll = ws[off[3]:off[3] + 4 * B].view(F32)
    return alpha, beta, nll


def libm_f32(x):
    """(expf(x), logf(x), the lattice's branch-free expf for x <= 0) from the device build of csrc/libm_f32.h."""
    _cc(x, "libm_f32.x")
    e, l, n = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    check(_L().dyn_libm_f32(x.data_ptr(), x.numel(), e.data_ptr(), l.data_ptr(), n.data_ptr(), _stream()), "dyn_libm_f32")
    return e, l, n


# ----------------------------------------------------------------------------------------------- optimiser / stitch
def madgrad_step(p, g, s, nu, x0, lr, momentum, weight_decay, eps, step):
    for t, n in ((p, "p"), (g, "g"), (s, "s"), (nu, "nu"), (x0, "x0")):
        _cc(t, "madgrad." + n)
    check(_L().dyn_madgrad_step(p.data_ptr(), g.data_ptr(), s.data_ptr(), nu.data_ptr(), x0.data_ptr(), p.numel(), lr, momentum,
                                weight_decay, eps, step, _stream()), "dyn_madgrad_step")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _cc(t, "adam." + n)
    check(_L().dyn_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps,
                             weight_decay, step, _stream()), "dyn_adam_step")


def clip_grad_norm(g, max_norm):
    """In-place global-norm clip of a flat gradient buffer; returns a 2-float device tensor (norm, coefficient)."""
    _cc(g, "clip_grad_norm.g")
    out = torch.empty(2, device=g.device, dtype=F32)
    ws = workspace(g.device)
    check(_L().dyn_clip_grad_norm(g.data_ptr(), g.numel(), max_norm, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
          "dyn_clip_grad_norm")
    return out


def stitch_accumulate(log_probs, acc, count, pos):
    _c(log_probs, "stitch.log_probs"); _cc(acc, "stitch.acc"); _cc(count, "stitch.count")
    rows, C = log_probs.shape
    check(_L().dyn_stitch_accumulate(log_probs.data_ptr(), C, acc.data_ptr(), count.data_ptr(), pos, rows, C, acc.shape[0],
                                     _stream()), "dyn_stitch_accumulate")


def stitch_finalize(acc, count, rows):
    _cc(acc, "stitch.acc"); _cc(count, "stitch.count")
    C = acc.shape[1]
    out = torch.empty(rows, C, device=acc.device, dtype=F32)
    check(_L().dyn_stitch_finalize(acc.data_ptr(), count.data_ptr(), out.data_ptr(), rows, C, _stream()), "dyn_stitch_finalize")
    return out


def stitch_finalize_rows(acc, count, row_index):
    """log(acc / count) over the listed rows (int64 CUDA tensor): coverage with gaps."""
    _cc(acc, "stitch.acc"); _cc(count, "stitch.count"); _c(row_index, "stitch.row_index", torch.int64)
    C = acc.shape[1]
    out = torch.empty(row_index.numel(), C, device=acc.device, dtype=F32)
    check(_L().dyn_stitch_finalize_rows(acc.data_ptr(), count.data_ptr(), row_index.data_ptr(), out.data_ptr(), row_index.numel(), C,
                                        _stream()), "dyn_stitch_finalize_rows")
    return out


# ----------------------------------------------------------------------------------------------- wav2vec2 pieces
def gelu(x, out=None):
    _c(x, "gelu.x")
    out = torch.empty_like(x) if out is None else out
    check(_L().dyn_gelu_fwd(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "dyn_gelu_fwd")
    return out


def gelu_bwd(x, dy, out=None):
    _c(x, "gelu_bwd.x"); _c(dy, "gelu_bwd.dy")
    out = torch.empty_like(x) if out is None else out
    check(_L().dyn_gelu_bwd(x.data_ptr(), dy.data_ptr(), out.data_ptr(), x.numel(), _stream()), "dyn_gelu_bwd")
    return out


def _valid_ptr(valid, what):
    """`valid`: None, or a one-element int32 CUDA tensor holding a row / column count that the kernel reads when it RUNS (so a captured launch
    follows later updates of the tensor)."""
    if valid is None:
        return None
    if not (isinstance(valid, torch.Tensor) and valid.is_cuda and valid.dtype == torch.int32 and valid.numel() == 1):
        raise DynError(f"{what}: the valid-length operand must be a one-element int32 CUDA tensor")
    return valid.data_ptr()


def colnorm(x, gamma, beta, eps=1e-5, valid=None):
    """x [B, T, C]: normalise over T per (b, c) (GroupNorm with groups == channels). Returns (y, mean, rstd).  `valid` (device int32 scalar):
    the statistics cover the first `valid` rows only (zero-padded bucket)."""
    _c(x, "colnorm.x")
    B, T, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(B, C, device=x.device, dtype=F32); rstd = torch.empty(B, C, device=x.device, dtype=F32)
    ws = workspace(x.device)
    check(_L().dyn_colnorm_fwd_len(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, T, C,
                                   eps, _valid_ptr(valid, "colnorm"), ws.data_ptr(), ws.numel(), _stream()), "dyn_colnorm_fwd")
    return y, mean, rstd


def colnorm_bwd(x, gamma, mean, rstd, dy, dgamma, dbeta, wgrad_beta=1.0, valid=None):
    B, T, C = x.shape
    dx = torch.empty_like(x)
    ws = workspace(x.device)
    check(_L().dyn_colnorm_bwd_len(x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dx.data_ptr(),
                                   _opt(dgamma, "dgamma"), _opt(dbeta, "dbeta"), wgrad_beta, B, T, C, _valid_ptr(valid, "colnorm_bwd"),
                                   ws.data_ptr(), ws.numel(), _stream()), "dyn_colnorm_bwd")
    return dx


def mask_rows(x, valid):
    """x [B, T, C] in place: rows t >= valid (device int32 scalar) of every batch entry := 0."""
    _cc(x, "mask_rows.x")
    B, T, C = x.shape
    check(_L().dyn_mask_rows(x.data_ptr(), B, T, C, _valid_ptr(valid, "mask_rows"), _stream()), "dyn_mask_rows")
    return x


def conv1d_out_len(T, kw, stride):
    return (T - kw) // stride + 1


def conv1d(x, w, kw, stride):
    """Valid strided Conv1d as an implicit GEMM: x [B, T, Cin] channels-last, w [Cout, kw*Cin] -> [B, Tout, Cout]."""
    _c(x, "conv1d.x"); _c(w, "conv1d.w")
    B, T, Cin = x.shape
    Cout = w.shape[0]
    Tout = conv1d_out_len(T, kw, stride)
    y = torch.empty(B, Tout, Cout, device=x.device, dtype=F32)
    gemm(x, w, y, trans_b=True, M=Tout, N=Cout, K=kw * Cin, lda=stride * Cin, ldb=kw * Cin, ldc=Cout, nb1=B,
         sa=(T * Cin, 0), sc=(Tout * Cout, 0))
    return y


def conv1d_wgrad(x, dy, dw, kw, stride, beta=1.0):
    """dw [Cout, kw*Cin] = beta*dw + sum_b dy[b]^T @ rows(x[b])  (rows overlap: ldb = stride*Cin < kw*Cin)."""
    B, T, Cin = x.shape
    Tout, Cout = dy.shape[1], dy.shape[2]
    for b in range(B):
        gemm(dy, x, dw, trans_a=True, M=Cout, N=kw * Cin, K=Tout, lda=Cout, ldb=stride * Cin, ldc=kw * Cin,
             a_off=b * Tout * Cout, b_off=b * T * Cin, beta=beta if b == 0 else 1.0)
    return dw


def conv1d_dgrad(dy, w, T, Cin, kw, stride):
    """dx [B, T, Cin] from dy [B, Tout, Cout]: dense row gradients (GEMM) then col2im over the overlapping rows."""
    B, Tout, Cout = dy.shape
    dA = torch.empty(B, Tout, kw * Cin, device=dy.device, dtype=F32)
    gemm(dy, w, dA, M=B * Tout, N=kw * Cin, K=Cout, lda=Cout, ldb=kw * Cin, ldc=kw * Cin)
    dx = torch.empty(B, T, Cin, device=dy.device, dtype=F32)
    check(_L().dyn_col2im_1d(dA.data_ptr(), dx.data_ptr(), B, T, Tout, Cin, kw, stride, _stream()), "dyn_col2im_1d")
    return dx


def weight_norm(v, g):
    """v [rows, kw, cg], g [kw] -> w = g[tap] * v / ||v[:, tap, :]||."""
    rows, kw, cg = v.shape
    w = torch.empty_like(v)
    ws = workspace(v.device)
    check(_L().dyn_weight_norm_fwd(v.data_ptr(), g.data_ptr(), w.data_ptr(), rows, kw, cg, ws.data_ptr(), ws.numel(), _stream()),
          "dyn_weight_norm_fwd")
    return w


def weight_norm_bwd(v, g, dw, dv, dg, beta=1.0):
    rows, kw, cg = v.shape
    ws = workspace(v.device)
    check(_L().dyn_weight_norm_bwd(v.data_ptr(), g.data_ptr(), dw.data_ptr(), dv.data_ptr(), dg.data_ptr(), beta, rows, kw, cg,
                                   ws.data_ptr(), ws.numel(), _stream()), "dyn_weight_norm_bwd")


def group_pack(x, G, pad):
    B, T, C = x.shape
    xg = torch.empty(B, G, T + 2 * pad, C // G, device=x.device, dtype=F32)
    check(_L().dyn_group_pack(x.data_ptr(), xg.data_ptr(), B, T, C, G, pad, _stream()), "dyn_group_pack")
    return xg


def group_unpack(yg, bias, T, C):
    B, G, Tg, cg = yg.shape
    y = torch.empty(B, T, C, device=yg.device, dtype=F32)
    check(_L().dyn_group_unpack(yg.data_ptr(), y.data_ptr(), _opt(bias, "bias"), B, T, Tg, C, G, _stream()), "dyn_group_unpack")
    return y


def group_pack_grad(dy, G, Tg):
    B, T, C = dy.shape
    dyg = torch.empty(B, G, Tg, C // G, device=dy.device, dtype=F32)
    check(_L().dyn_group_pack_grad(dy.data_ptr(), dyg.data_ptr(), B, T, Tg, C, G, _stream()), "dyn_group_pack_grad")
    return dyg


def group_unpack_grad(dxg, dx, pad, beta=0.0):
    B, T, C = dx.shape
    G = dxg.shape[1]
    check(_L().dyn_group_unpack_grad(dxg.data_ptr(), dx.data_ptr(), B, T, C, G, pad, beta, _stream()), "dyn_group_unpack_grad")
    return dx
