"""(Needs a diagnostic kernel build: commit 8bcdd1b for DYN_GEMM_DEBUG = 1 / 2, the commit named in DESIGN.md §6 for 3 / 4; the switches were
removed from the product source after the measurement.)
Diagnostic: NT GEMM (M = N = 4096, K = 768 / 3072) with a forced tile (argv[1], default 64) under DYN_GEMM_DEBUG (set in the environment of the
process: the switch is read once).  1 = no direct-to-LDS loads in the K loop, 2 = no workgroup barrier in the K loop, 3 = every workgroup
fetches the operands of tile (0, 0) (48 - 196 KB footprint: every fetch an L2 hit), 4 = of one of 8 tiles (results are garbage)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd import ops
dev = torch.device("cuda:0")


def warm(seconds=6.0):
    import time
    a = torch.randn(8192, 3072, device=dev); b = torch.randn(4096, 3072, device=dev); c = torch.empty(8192, 4096, device=dev)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            ops.gemm(a, b, c, trans_b=True, M=8192, N=4096, K=3072, lda=3072, ldb=3072, ldc=4096, force=(128, 128, 1))
        torch.cuda.synchronize()


warm()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for K in (768, 3072):
    M, N = 4096, 4096
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
    f = lambda: ops.gemm(a, b, c, trans_b=True, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, force=(T, T, 1))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"DYN_GEMM_DEBUG={os.environ.get('DYN_GEMM_DEBUG', '0')} NT {T}x{T} M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF/s (MFMA-equivalent)", flush=True)
