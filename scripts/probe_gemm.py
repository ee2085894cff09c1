"""Times the fp32 MFMA GEMM at the encoder's shapes (HIP events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd import ops

dev = torch.device("cuda:0")
shapes = [  # (name, ta, tb, M, N, K)
    ("ffn_up   NT", False, True, 4096, 3072, 768),
    ("ffn_down NT", False, True, 4096, 768, 3072),
    ("qkv      NT", False, True, 4096, 2304, 768),
    ("head     NT", False, True, 4096, 4096, 768),
    ("dgrad    NN", False, False, 4096, 768, 3072),
    ("wgrad    TN", True, False, 3072, 768, 4096),
    ("wgrad2   TN", True, False, 768, 768, 4096),
    ("square   NT", False, True, 4096, 4096, 4096),
    ("sub_pw   NT", False, True, 163840, 256, 256),
]
for name, ta, tb, M, N, K in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    c = torch.empty(M, N, device=dev)
    f = lambda: ops.gemm(a, b, c, trans_a=ta, trans_b=tb, M=M, N=N, K=K, lda=a.shape[1], ldb=b.shape[1], ldc=N)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    ref = torch.matmul(a.T if ta else a, b.T if tb else b)
    err = (ref - c).abs().max().item()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): torch.matmul(a.T if ta else a, b.T if tb else b)
    t1.record(); torch.cuda.synchronize()
    ms_t = t0.elapsed_time(t1) / n
    print(f"{name}: M={M} N={N} K={K}  {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s  (hipBLAS {2*M*N*K/ms_t/1e9:.1f})  maxdiff_vs_torch={err:.2e}", flush=True)
