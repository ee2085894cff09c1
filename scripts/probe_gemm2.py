"""GEMM probe for arbitrary NT shapes: python scripts/probe_gemm2.py M,N,K [M,N,K ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd import ops
dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    parts = spec.split(",")
    M, N, K = map(int, parts[:3])
    mode = parts[3] if len(parts) > 3 else "NT"
    force = tuple(int(v) for v in parts[4].split(":")) if len(parts) > 4 else None     # bm:bn:tail_slices
    ta, tb = mode[0] == "T", mode[1] == "T"
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev)
    c = torch.empty(M, N, device=dev)
    f = lambda: ops.gemm(a, b, c, trans_a=ta, trans_b=tb, M=M, N=N, K=K, lda=a.shape[1], ldb=b.shape[1], ldc=N, force=force)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{mode} M={M} N={N} K={K} {force or ''}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:6.1f} TF/s", flush=True)
