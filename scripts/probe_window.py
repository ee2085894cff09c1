"""Per-shape GEMM time inside the real dynamic-eval step (every launch timed with HIP events)."""
import sys, os, argparse, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd import ops, lib
from dynamic_asr_eval_amd.model import SCConformerXL
from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
from dynamic_asr_eval_amd.datasets import synthetic_spec
from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer

dev = torch.device("cuda:0")
model = SCConformerXL(vocab_size=4095, device=dev)
init_synthetic(model, seed=0, blank_bias=1.34)
tok = SyntheticTokenizer(4095)
args = argparse.Namespace(config={'model': {'subsampling_factor': 8}, 'audio_chunking': {'size': 16384, 'overlap': 0}, 'training': {}},
                          optim_lr=9e-5, epochs=1, shuffle=False, quiet=True, spec_augment_n_freq_masks=6, spec_augment_freq_mask_param=34)
spec = synthetic_spec(16384 + 3 * 2048, seed=1).to(dev)
lib.dynamic_eval(args, model, spec, 16384, 14336, tok, use_tqdm=False, return_device=True)  # warm-up
shapes = []
orig = ops.gemm
def timed(a, b, c, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(a, b, c, **kw); e1.record()
    shapes.append(((kw.get('trans_a', False), kw.get('trans_b', False), kw['M'], kw['N'], kw['K'], kw.get('nb1', 1) * kw.get('nb2', 1)), e0, e1))
    return r
ops.gemm = timed
torch.cuda.synchronize()
import time
t0 = time.time()
lib.dynamic_eval(args, model, spec, 16384, 14336, tok, use_tqdm=False, return_device=True)
torch.cuda.synchronize()
wall = time.time() - t0
agg = collections.defaultdict(lambda: [0, 0.0])
for k, e0, e1 in shapes:
    agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"wall {wall*1e3:.1f} ms, gemm total {tot:.1f} ms over {len(shapes)} launches")
for k, (n, ms) in sorted(agg.items(), key=lambda x: -x[1][1])[:40]:
    ta, tb, M, N, K, nb = k
    fl = 2.0 * M * N * K * nb * n
    print(f"{'T' if ta else 'N'}{'T' if tb else 'N'} M={M:6d} N={N:5d} K={K:5d} nb={nb:3d} calls={n:4d} {ms:8.2f} ms {ms/tot*100:5.1f}% avg={ms/n*1e3:7.1f}us {fl/ms/1e9:6.1f} TF/s")
