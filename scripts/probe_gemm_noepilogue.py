"""(Diagnostic build only: epilogue codes 77 = return before the stores, 78 = every tile stores into the first tile (L2 hits), 79 = nt stores, 80 = sc1 write-through stores.)  128x128 NT launches with and without the epilogue: is the ~30 us per
launch at K = 768 the exposed last epilogue?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynamic_asr_eval_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
N, K = 4096, 768
for epi in (0, 77, 78, 79, 80):
    pts = []
    for rounds in (1, 2, 3, 4, 8):
        M = int(rounds * 512) // (N // 128) * 128
        a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); c = torch.empty(M, N, device=dev)
        us = timeit(lambda: ops.gemm(a, b, c, trans_b=True, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, force=(128, 128, 1), epilogue=epi))
        pts.append((rounds, us)); print(f"epilogue={epi} 128x128 K={K} {rounds} rounds: {us:7.1f} us", flush=True)
    r = np.array([p[0] for p in pts]); t = np.array([p[1] for p in pts]); bfit, afit = np.polyfit(r, t, 1)
    print(f"  epilogue={epi}: {afit:5.1f} us per launch + {bfit:5.1f} us per round", flush=True)
