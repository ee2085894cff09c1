"""Calibrates the synthetic model's blank bias: fraction of non-blank frames / tokens per 16384-frame window."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd.model import SCConformerXL
from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
from dynamic_asr_eval_amd.datasets import synthetic_spec
from dynamic_asr_eval_amd import ops
dev = torch.device("cuda:0")
m = SCConformerXL(vocab_size=4095, device=dev)
spec = synthetic_spec(16384, seed=1234).to(dev)
for bb in (1.1, 1.2, 1.25, 1.3, 1.35, 1.4):
    init_synthetic(m, seed=0, blank_bias=bb)
    with torch.no_grad():
        lp = m(audio_signal=spec)['final_posteriors']
    ids, n = ops.ctc_greedy(lp, 4095)
    nb = (lp[0].argmax(-1) != 4095).float().mean().item()
    print(f"blank_bias={bb}: non-blank frames {nb:.3f}, tokens {n.item()} of {lp.shape[1]} frames, max logp mean {lp[0].max(-1).values.mean().item():.3f}", flush=True)
