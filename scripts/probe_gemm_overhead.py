"""Per-launch overhead of dyn_gemm_f32 against work per launch: NT products with a forced tile, N = 4096, M chosen so that the launch is
0.5 .. 8 rounds of resident workgroups (slots = 2 per CU for 128x128, 4 for 64x64), K = 768 and 3072.  A fit time = a + b * rounds
separates the steady-state rate (b: one round of tiles) from what a launch pays once (a: ramp, first-round lockstep, tail)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dynamic_asr_eval_amd import ops

dev = torch.device("cuda:0")


def warm_gpu(seconds=6.0):
    """The first seconds of a process on an idle MI355X run slower (clock ramp): whatever is measured first reads low.  Load the chip first."""
    import time
    a = torch.randn(8192, 3072, device=dev); b = torch.randn(4096, 3072, device=dev); c = torch.empty(8192, 4096, device=dev)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            ops.gemm(a, b, c, trans_b=True, M=8192, N=4096, K=3072, lda=3072, ldb=3072, ldc=4096)
        torch.cuda.synchronize()


warm_gpu()


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


N = 4096
for (bm, bn, slots) in ((128, 128, 512), (64, 64, 1024), (64, 128, 768)):
    for K in (768, 3072):
        pts = []
        for rounds in (0.5, 1, 2, 3, 4, 8):
            tiles = int(rounds * slots)
            M = tiles // (N // bn) * bm
            a = torch.randn(M, K, device=dev)
            b = torch.randn(N, K, device=dev)
            c = torch.empty(M, N, device=dev)
            us = timeit(lambda: ops.gemm(a, b, c, trans_b=True, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, force=(bm, bn, 1)))
            pts.append((rounds, us))
            print(f"tile {bm}x{bn} K={K} M={M:6d} ({rounds:3.1f} rounds): {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF/s", flush=True)
            del a, b, c
        r = np.array([p[0] for p in pts if p[0] >= 1]); t = np.array([p[1] for p in pts if p[0] >= 1])
        bfit, afit = np.polyfit(r, t, 1)
        ideal = 2.0 * bm * bn * K * slots / 157.3e6
        print(f"  fit (rounds >= 1): {afit:6.1f} us per launch + {bfit:6.1f} us per round (one round at the fp32-MFMA peak: {ideal:.1f} us)", flush=True)
