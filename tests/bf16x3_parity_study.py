"""CPU-only study (lives under tests/ because it imports the oracle): would the dynamic-eval loop stay inside its parity bars if the matrix
products were computed as bf16x3 SPLIT PRODUCTS (every fp32 operand = three bf16 terms, six bf16 x bf16 products per fp32 product, fp32
accumulation: DESIGN.md section 6, scripts/probe_bf16x3_numerics.py) instead of fp32 FMA chains?  The CPU oracle (oracle/conformer_ref.py, the
model of BASELINE config 2: 6 x 768, V + 1 = 4096) runs the first windows of a recording three ways — float32, float32 with every
F.linear / matmul (forward AND backward) replaced by the bf16x3 emulation, and float64 — with the same weights, SpecAugment masks and MADGRAD
steps, and the stitched log-probs are compared: if |bf16x3 - f64| is of the size of |f32 - f64|, the split products are fp32-grade for this loop.
    python tests/bf16x3_parity_study.py [--windows 3] [--out profiles/r04_bf16x3_loop_study.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VOCAB, SEQ, OVL = 4095, 16384, 14336
_matmul = torch.matmul


def split3(x):
    a0 = x.to(torch.bfloat16).to(torch.float32)
    r = x - a0
    a1 = r.to(torch.bfloat16).to(torch.float32)
    r = r - a1
    return a0, a1, r.to(torch.bfloat16).to(torch.float32)


def mm3(a, b):
    """a @ b with both operands split into three bf16 terms, the six leading products (each exact in fp32) accumulated in float32."""
    A, B = split3(a), split3(b)
    small = (_matmul(A[0], B[2]) + _matmul(A[2], B[0])) + _matmul(A[1], B[1])
    mid = _matmul(A[0], B[1]) + _matmul(A[1], B[0])
    return (small + mid) + _matmul(A[0], B[0])


class MatMul3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return mm3(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return mm3(g, b.transpose(-1, -2)), mm3(a.transpose(-1, -2), g)


class Linear3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        y = mm3(x.reshape(-1, x.shape[-1]), w.t()).reshape(*x.shape[:-1], w.shape[0])
        return y + bias if bias is not None else y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        return mm3(g2, w).reshape(x.shape), mm3(g2.t(), x2), (g2.sum(0) if ctx.has_bias else None)


class bf16x3_products:
    """While active, F.linear and the `@` operator of float32 tensors go through the split-product emulation (float64 tensors are left alone)."""

    def __enter__(self):
        self.lin, self.mm = F.linear, torch.Tensor.__matmul__
        F.linear = lambda x, w, b=None: Linear3.apply(x, w, b) if x.dtype == torch.float32 else self.lin(x, w, b)
        torch.Tensor.__matmul__ = lambda a, b: MatMul3.apply(a, b) if a.dtype == torch.float32 and a.dim() >= 2 and b.dim() >= 2 else self.mm(a, b)

    def __exit__(self, *exc):
        F.linear, torch.Tensor.__matmul__ = self.lin, self.mm
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=3)
    ap.add_argument("--lr", type=float, default=9e-5)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--blank_bias", type=float, default=1.34)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_bf16x3_loop_study.json"))
    a = ap.parse_args()
    from oracle import dynamic_eval_ref as R
    from oracle.conformer_ref import SCConformerXLRef
    from oracle.madgrad_ref import MADGRAD as MADGRAD_REF
    from dynamic_asr_eval_amd.datasets import synthetic_spec
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    torch.set_num_threads(a.threads)
    # a sanity check of the emulation itself against float64 on one product
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(256, 768, generator=g), torch.randn(768, 768, generator=g) * 0.05
    ref = x.double() @ w.double()
    e3, e1 = (mm3(x, w).double() - ref).abs().max().item(), ((x @ w).double() - ref).abs().max().item()
    print(f"[study] one 256x768x768 product: |bf16x3 - f64| {e3:.2e}, |MKL f32 - f64| {e1:.2e}", flush=True)
    tok = SyntheticTokenizer(VOCAB)
    spec = synthetic_spec(SEQ + (a.windows - 1) * (SEQ - OVL), seed=77)
    _, keys = R.prepare_chunks(spec, SEQ, OVL)
    gm = torch.Generator().manual_seed(9)
    masks = {k: (R.draw_masks(6, 34, 80, gm), ([], [])) for k in keys}
    res = {"what": __doc__.split("\n\n")[0], "windows": len(keys), "frames": int(spec.shape[-1]), "lr": a.lr, "host_threads": a.threads,
           "one_product_256x768x768": {"bf16x3_vs_f64": e3, "mkl_f32_vs_f64": e1}}

    def run(tag, dtype, emulate):
        model = SCConformerXLRef(vocab_size=VOCAB, seed=0, blank_bias=a.blank_bias)
        sp = spec
        if dtype == torch.float64:
            model, sp = model.double(), spec.double()
        t0 = time.time()
        if emulate:
            with bf16x3_products():
                off, on = R.dynamic_eval_ref(model, sp, SEQ, OVL, tok, MADGRAD_REF, {'lr': a.lr}, {}, fixed_masks=masks, also_online=True)
        else:
            off, on = R.dynamic_eval_ref(model, sp, SEQ, OVL, tok, MADGRAD_REF, {'lr': a.lr}, {}, fixed_masks=masks, also_online=True)
        print(f"[study] {tag}: {time.time() - t0:.0f} s", flush=True)
        return np.asarray(off, dtype=np.float64), np.asarray(on, dtype=np.float64)
    f32 = run("float32 oracle", torch.float32, False)
    b3 = run("float32 oracle with bf16x3 split products", torch.float32, True)
    f64 = run("float64 oracle", torch.float64, False)

    def cmp(x, y):
        return {"offline_max_abs_dlogp": float(np.abs(x[0] - y[0]).max()), "online_max_abs_dlogp": float(np.abs(x[1] - y[1]).max()),
                "argmax_mismatch_frames": [int((x[0].argmax(-1) != y[0].argmax(-1)).sum()), int((x[1].argmax(-1) != y[1].argmax(-1)).sum())]}
    res["f32_vs_f64"], res["bf16x3_vs_f64"], res["bf16x3_vs_f32"] = cmp(f32, f64), cmp(b3, f64), cmp(b3, f32)
    res["frames_compared"] = int(f32[0].shape[0])
    print(json.dumps(res, indent=1))
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
