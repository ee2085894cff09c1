"""Seeded synthetic weights for benchmarks (no checkpoint of the acoustic model exists offline).
Rule (identical to oracle/conformer_ref.init_parameters, so both sides build the same model from a seed):
one torch CPU generator walked in named_parameters order; matrices / conv kernels ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in));
biases ~ U(-0.05, 0.05); norm gains 1 + U(-0.05, 0.05); `blank_bias` is added to the CTC head's blank logit so the
greedy pseudo-labels have a speech-like token rate instead of one token per frame."""
import math

import torch


def init_synthetic(model, seed=0, blank_bias=0.0):
    g = torch.Generator().manual_seed(int(seed))
    for name, p in model.named_parameters():
        if name.endswith("norm.weight") or name.endswith("norm_out.weight") or name.endswith("cnorm.weight"):
            v = 1.0 + 0.1 * (torch.rand(p.shape, generator=g) - 0.5)
        elif p.dim() == 1:
            v = 0.1 * (torch.rand(p.shape, generator=g) - 0.5)
        else:
            fan_in = math.prod(p.shape[1:])
            v = (torch.rand(p.shape, generator=g) * 2 - 1) * (1.0 / math.sqrt(fan_in))
        if blank_bias and name == "decoder.ff.bias":
            v[-1] += blank_bias
        p.copy_(v.to(p.device))
    return model


def calibrate_blank_bias(model, window, lo=0.0, hi=4.0, target=(250, 700), iters=14):
    """Bisection on the CTC head's blank logit bias so that the greedy transcript of `window` ([1, F, T] on the GPU) has
    a speech-like length (`target` tokens, ~1.5-4 tokens/s for a 164 s window) instead of the degenerate all-token /
    all-blank output of a randomly initialised encoder.  Synthetic-weights shaping only; returns the bias used."""
    import torch
    from . import ops
    bias = model.P["decoder.ff.bias"]
    blank = model.num_classes - 1
    base = float(bias[-1].item())
    mid = 0.5 * (lo + hi)
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        bias[-1] = base + mid
        with torch.no_grad():
            lp = model(audio_signal=window)['final_posteriors']
        _, n = ops.ctc_greedy(lp, blank)
        n = int(n[0].item())
        nonblank = float((lp[0].argmax(-1) != blank).float().mean().item())
        if target[0] <= n <= target[1]:
            break
        if nonblank > 0.5:   # too many token frames -> raise the blank
            lo = mid
        else:
            hi = mid
    return mid
