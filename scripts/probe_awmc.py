"""AWMC (reference lcasr/lib.py:206-376) throughput on one 20-min synthetic recording, eager and with hipGraph replay."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd import lib
from dynamic_asr_eval_amd.model import SCConformerXL
from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
from dynamic_asr_eval_amd.datasets import synthetic_spec
from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
dev = torch.device("cuda:0")
model = SCConformerXL(vocab_size=4095, device=dev); init_synthetic(model, seed=0, blank_bias=1.34)
tok = SyntheticTokenizer(4095)
secs = 1200.0
spec = synthetic_spec(int(secs * 100), seed=2).to(dev)
for graphs in (False, True):
    a = argparse.Namespace(config={'model': {'subsampling_factor': 8}, 'audio_chunking': {'size': 16384, 'overlap': 0}, 'training': {}},
                           optim_lr=9e-5, epochs=1, shuffle=False, quiet=True, use_graphs=graphs, spec_augment_n_freq_masks=6, spec_augment_freq_mask_param=34)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lib.AWMC(a, model, spec, 16384, 14336, tok, use_tqdm=False, return_device=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"AWMC graphs={graphs}: {dt:.2f} s for {secs:.0f} s of audio = {secs/dt:.1f} audio-s/s (RTF {dt/secs:.4f})", flush=True)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lib.dynamic_eval(a, model, spec, 16384, 14336, tok, use_tqdm=False, return_device=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"dynamic_eval graphs={graphs}: {secs/dt:.1f} audio-s/s (RTF {dt/secs:.4f})", flush=True)
