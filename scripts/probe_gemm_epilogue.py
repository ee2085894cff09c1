"""Cost of the GEMM epilogue variants on the model's narrow-output shapes: plain store vs bias vs residual read (beta = 1, C_in) —
the tuned plan of each shape.  usage: python scripts/probe_gemm_epilogue.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd import ops
dev = torch.device("cuda:0")


def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in ((4096, 768, 3072), (4096, 768, 768), (4096, 3072, 768), (8192, 768, 3072), (2048, 768, 3072)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev); out = torch.empty(M, N, device=dev)
    t_plain = timeit(lambda: ops.linear(x, w, out=out))
    t_bias = timeit(lambda: ops.linear(x, w, bias, out=out))
    t_res = timeit(lambda: ops.linear(x, w, bias, out=out, beta=1.0, residual=res))
    t_inpl = timeit(lambda: ops.linear(x, w, bias, out=out, beta=1.0))
    fl = 2.0 * M * N * K
    print(f"NT M={M} N={N} K={K}: plain {t_plain:7.1f} us ({fl/t_plain/1e6:5.1f} TF/s) | +bias {t_bias:7.1f} | +bias +residual(other buffer) {t_res:7.1f} ({fl/t_res/1e6:5.1f} TF/s) | +bias, beta=1 in place {t_inpl:7.1f}", flush=True)
