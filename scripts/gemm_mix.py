"""Which GEMM shapes the adapt step spends its time in: counts every dyn_gemm_f32 call of one full window step (B=2 forward,
B=1 backward, final-pass forward batch of 4) and times each distinct shape in isolation on the real operands."""
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
                          optim_lr=9e-5, epochs=1, shuffle=False, quiet=True, use_graphs=False, spec_augment_n_freq_masks=6, spec_augment_freq_mask_param=34)
spec = synthetic_spec(16384 + 7 * 2048, seed=1).to(dev)     # 8 full windows: 8 adapt steps + 2 final-pass batches of 4
orig = ops.gemm
stat = collections.OrderedDict()

def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def wrapped(a, b, c, **kw):
    key = ("T" if kw.get('trans_a') else "N") + ("T" if kw.get('trans_b') else "N"), kw['M'], kw['N'], kw['K'], kw.get('nb1', 1) * kw.get('nb2', 1)
    if key not in stat:
        scratch = c.clone()
        stat[key] = [0, timeit(lambda: orig(a, b, scratch, **kw))]
    stat[key][0] += 1
    return orig(a, b, c, **kw)

ops.gemm = wrapped
lib.dynamic_eval(args, model, spec, 16384, 14336, tok, use_tqdm=False, return_device=True)
torch.cuda.synchronize()
rows = []
for (mode, M, N, K, nb), (cnt, ms) in stat.items():
    fl = 2.0 * M * N * K * nb
    rows.append((cnt * ms, mode, M, N, K, nb, cnt, ms, fl / ms / 1e9))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows); totfl = sum(2.0 * r[2] * r[3] * r[4] * r[5] * r[6] for r in rows)
print(f"total GEMM time {tot:.1f} ms over 8 windows, {totfl/tot/1e9:.1f} TF/s aggregate")
acc = 0
for t, mode, M, N, K, nb, cnt, ms, tf in rows[:40]:
    acc += t
    print(f"{mode} M={M:6d} N={N:5d} K={K:5d} nb={nb:2d}  x{cnt:4d}  {ms*1e3:7.1f} us  {tf:6.1f} TF/s  {100*t/tot:5.1f}%  cum {100*acc/tot:5.1f}%")
