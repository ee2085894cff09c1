"""Aggregate GEMM throughput with 1..4 HIP streams each looping over the adapt step's typical shapes (own operands and scratch per stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd import ops

dev = torch.device("cuda:0")
SHAPES = [("NT", 4096, 3072, 768), ("NT", 4096, 768, 3072), ("NT", 4096, 2304, 768), ("NT", 4096, 768, 768), ("NT", 4096, 4096, 768),
          ("NN", 2048, 768, 3072), ("NN", 2048, 3072, 768), ("NN", 2048, 768, 768), ("NN", 2048, 768, 4096), ("NT", 4096, 1536, 768)]


def make(stream_idx):
    items = []
    for mode, M, N, K in SHAPES:
        ta, tb = mode[0] == "T", mode[1] == "T"
        a = torch.randn((K, M) if ta else (M, K), device=dev)
        b = torch.randn((N, K) if tb else (K, N), device=dev)
        c = torch.empty(M, N, device=dev)
        items.append((a, b, c, dict(trans_a=ta, trans_b=tb, M=M, N=N, K=K, lda=a.shape[1], ldb=b.shape[1], ldc=N)))
    ws = torch.empty(ops.WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
    return items, ws


flops_per_pass = sum(2.0 * M * N * K for _, M, N, K in SHAPES)
for n_streams in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    work = [make(i) for i in range(n_streams)]
    graphs = []
    for st, (items, ws) in zip(streams, work):       # one hipGraph per stream: 10 passes over the shape list (no host launch cost in the loop)
        with torch.cuda.stream(st):
            with ops.use_workspace(ws):
                for a, b, c, kw in items:
                    ops.gemm(a, b, c, **kw)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                with ops.use_workspace(ws):
                    for _ in range(10):
                        for a, b, c, kw in items:
                            ops.gemm(a, b, c, **kw)
            graphs.append(g)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(6):
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    total = flops_per_pass * 10 * 6 * n_streams
    print(f"{n_streams} stream(s): {total / dt / 1e12:6.1f} TFLOP/s aggregate ({dt * 1e3 / (60 * len(SHAPES)):.1f} us per GEMM per stream)", flush=True)
