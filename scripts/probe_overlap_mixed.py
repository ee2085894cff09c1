"""Can an HBM-bound kernel stream and a GEMM stream share the chip?  Stream A replays a hipGraph of the adapt step's GEMM shapes, stream B a hipGraph
of HBM-bound launches (LayerNorm over [4096, 768], SiLU over [4096, 3072], axpby over 86M floats = the optimiser's footprint), sized to take about as
long alone.  Prints each alone, both together, and the overlap gain = (t_A + t_B) / t_both (1.0 = the chip serialises them, 2.0 = one hides fully)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd import ops

dev = torch.device("cuda:0")
SHAPES = [("NT", 4096, 3072, 768), ("NT", 4096, 768, 3072), ("NT", 4096, 2304, 768), ("NT", 4096, 768, 768), ("NT", 4096, 4096, 768),
          ("NN", 2048, 768, 3072), ("NN", 2048, 3072, 768), ("NN", 2048, 768, 768), ("NN", 2048, 768, 4096), ("NT", 4096, 1536, 768)]
PRIO = os.environ.get('PROBE_PRIO', '0')        # 'hbm': the HBM stream is high priority, 'gemm': the GEMM stream is
sa = torch.cuda.Stream(priority=-1 if PRIO == 'gemm' else 0)
sb = torch.cuda.Stream(priority=-1 if PRIO == 'hbm' else 0)
print('priority:', PRIO, torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else '')


def gemm_graph(passes):
    items = []
    for mode, M, N, K in SHAPES:
        ta, tb = mode[0] == "T", mode[1] == "T"
        a = torch.randn((K, M) if ta else (M, K), device=dev)
        b = torch.randn((N, K) if tb else (K, N), device=dev)
        items.append((a, b, torch.empty(M, N, device=dev), dict(trans_a=ta, trans_b=tb, M=M, N=N, K=K, lda=a.shape[1], ldb=b.shape[1], ldc=N)))
    ws = torch.empty(ops.WORKSPACE_BYTES, dtype=torch.uint8, device=dev)

    def body():
        with ops.use_workspace(ws):
            for a, b, c, kw in items:
                ops.gemm(a, b, c, **kw)
    with torch.cuda.stream(sa):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=sa):
            for _ in range(passes):
                body()
    return g, (items, ws)


def hbm_graph(passes, kind):
    x = torch.randn(4096, 768, device=dev); gam = torch.ones(768, device=dev); bet = torch.zeros(768, device=dev); y = torch.empty_like(x)
    u = torch.randn(4096, 3072, device=dev); v = torch.empty_like(u)
    p = torch.randn(86_000_000, device=dev); q = torch.randn(86_000_000, device=dev)

    def body():
        if kind in ("small", "mixed"):
            for _ in range(6):
                ops.layernorm(x, gam, bet, 1e-5, out=y)
                ops.silu(u, out=v)
        if kind in ("big", "mixed"):
            ops.axpby(p, q, 0.5, 0.5)
    with torch.cuda.stream(sb):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=sb):
            for _ in range(passes):
                body()
    return g, (x, gam, bet, y, u, v, p, q)


def run(ga, gb, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if ga is not None:
            with torch.cuda.stream(sa):
                ga.replay()
        if gb is not None:
            with torch.cuda.stream(sb):
                gb.replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


ga, keep_a = gemm_graph(20)
t0 = time.time()
while time.time() - t0 < 6.0:           # warm the chip
    run(ga, None, 1)
ta = run(ga, None)
for kind in ("small", "big", "mixed"):
    gb1, keep = hbm_graph(1, kind)
    t1 = run(None, gb1)
    passes = max(1, int(round(ta / t1)))
    del gb1, keep
    gb, keep = hbm_graph(passes, kind)
    tb = run(None, gb)
    both = run(ga, gb)
    print(f"{kind:6s}: GEMM stream alone {ta * 1e3:7.2f} ms, HBM stream alone {tb * 1e3:7.2f} ms ({passes} passes), together {both * 1e3:7.2f} ms -> "
          f"overlap gain {(ta + tb) / both:.3f} (serial = 1.0)", flush=True)
    del gb, keep
