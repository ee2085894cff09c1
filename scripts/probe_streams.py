"""One host thread, two HIP streams, two model replicas: eager vs hipGraph replay, sequential vs interleaved."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd.model import SCConformerXL
from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
dev = torch.device("cuda:0")
models = [SCConformerXL(vocab_size=4095, device=dev) for _ in range(2)]
for m in models: init_synthetic(m, 0)
x = torch.randn(2, 80, 16384, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
g = None
def step(m):
    global g
    out = m(audio_signal=x)['final_posteriors']
    if g is None: g = torch.zeros_like(out[:1])
    m.zero_grad(); m.backward(g, n_active=1)
for graphs in (False, True):
    for m in models: m.use_graphs = graphs
    for s, m in zip(streams, models):
        with torch.cuda.stream(s):
            for _ in range(3): step(m)
    torch.cuda.synchronize()
    N = 6
    t = time.time()
    for _ in range(N):
        for m in models:
            with torch.cuda.stream(streams[0]): step(m)
    host_seq = time.time() - t
    torch.cuda.synchronize(); seq = time.time() - t
    t = time.time()
    for _ in range(N):
        for s, m in zip(streams, models):
            with torch.cuda.stream(s): step(m)
    host_par = time.time() - t
    torch.cuda.synchronize(); par = time.time() - t
    print(f"graphs={graphs}: one stream {seq/N*1e3:.1f} ms per pair (host enqueue {host_seq/N*1e3:.1f}), two streams {par/N*1e3:.1f} ms per pair (host {host_par/N*1e3:.1f})", flush=True)
