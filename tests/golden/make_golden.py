"""Generates the committed fixtures under tests/golden/ (run once in the build container; the GPU box never sees
/root/reference).

  prepare_chunks.json   — key lists produced by the REFERENCE's own `prepare_chunks`: the function's source is pulled
                          out of /root/reference/lcasr/lib.py with `ast` (the module itself cannot be imported:
                          omegaconf/lcasr/... are absent) and executed as is; it is pure Python over tensor shapes.
  stitch_toy.json       — coverage counts of a 3-window toy worked by hand from reference lcasr/lib.py:615-629.
  tokenizer_128.model   — data asset copied from /root/reference/lcasr_nemo/tokenizer.model (SentencePiece, 128 pieces):
                          the only tokenizer the reference ships.
  speaker_manifest_15x15.json — data asset copied from /root/reference/lcasr/results/gender_eval_tedlium/ (talk ids per gender).
  softdtw_17x15x2.npz   — soft-DTW value/gradient at the reference's first self-check shape (soft_dtw_cuda.py:426),
                          produced by oracle/softdtw_ref.py (the reference's numba code cannot run here).
"""
import ast
import json
import os
import shutil
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def reference_prepare_chunks():
    src = open(os.path.join(REF, "lcasr", "lib.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "prepare_chunks"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "reference:lcasr/lib.py:prepare_chunks", "exec"), ns)
    return ns["prepare_chunks"]


def kernel_fixtures():
    """Vectors from the ops the reference itself calls (torch CPU), so the HIP kernels are pinned to committed numbers:
      ctc_64x10x129.npz   torch.nn.CTCLoss(blank=128, reduction='sum') loss + gradient w.r.t. log-probs (reference lcasr/lib.py:492,575,579)
      adam_3step.npz      torch.optim.Adam trajectory (reference nvidia_ctc/lib.py:43,155-160)
      madgrad_3step.npz   oracle/madgrad_ref.py trajectory (MADGRAD is un-vendored: parity unpinned, drift guard only)
      dyneval_trace.npz   2-layer model, 1100-frame recording, 512/256 windows: the whole dynamic-eval loop by the oracle with the
                          SpecAugment masks stored (RNG does not enter parity): stitched log-probs, argmax ids, adapted-parameter checksum."""
    g = torch.Generator().manual_seed(4242)
    T, B, C, S = 64, 2, 129, 10
    lp = torch.log_softmax(torch.randn(T, B, C, generator=g), -1).requires_grad_(True)
    tg = torch.randint(0, 128, (B, S), generator=g)
    tg[0, 3] = tg[0, 2]                                   # a repeated label (needs the blank between)
    il, tl = torch.tensor([T, T - 9]), torch.tensor([S, S - 3])
    loss = torch.nn.CTCLoss(blank=128, reduction='sum')(lp, tg, il, tl)
    loss.backward()
    np.savez(os.path.join(HERE, "ctc_64x10x129.npz"), log_probs=lp.detach().numpy(), targets=tg.numpy().astype(np.int32),
             input_lengths=il.numpy().astype(np.int32), target_lengths=tl.numpy().astype(np.int32), loss=float(loss), grad=lp.grad.numpy())

    p0 = torch.randn(1000, generator=g)
    grads = [torch.randn(1000, generator=g) * (0.5 + k) for k in range(3)]
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p], lr=1e-3)
    traj = []
    for gr in grads:
        p.grad = gr.clone(); opt.step(); traj.append(p.detach().clone().numpy())
    np.savez(os.path.join(HERE, "adam_3step.npz"), p0=p0.numpy(), grads=np.stack([x.numpy() for x in grads]), params=np.stack(traj), lr=1e-3)
    from oracle.madgrad_ref import MADGRAD
    p = torch.nn.Parameter(p0.clone())
    opt = MADGRAD([p], lr=1e-2)
    traj = []
    for gr in grads:
        p.grad = gr.clone(); opt.step(); traj.append(p.detach().clone().numpy())
    np.savez(os.path.join(HERE, "madgrad_3step.npz"), p0=p0.numpy(), grads=np.stack([x.numpy() for x in grads]), params=np.stack(traj), lr=1e-2)

    from oracle import dynamic_eval_ref as R
    from oracle.conformer_ref import SCConformerXLRef
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    cfg = dict(n_layers=2, d_model=256, n_heads=2, head_dim=128, subsampling_conv_channels=64)
    ref = SCConformerXLRef(cfg, vocab_size=128, seed=5, blank_bias=1.5)
    spec = torch.randn(1, 80, 1100, generator=g)
    _, keys = R.prepare_chunks(spec, 512, 256)
    mg = torch.Generator().manual_seed(7)
    masks = {k: (R.draw_masks(3, 12, 80, mg), ([], [])) for k in keys}
    out, params = R.dynamic_eval_ref(ref, spec, 512, 256, SyntheticTokenizer(128), MADGRAD, {'lr': 1e-4}, {}, epochs=1, shuffle=False,
                                     online=False, fixed_masks=masks, return_params=True)
    np.savez(os.path.join(HERE, "dyneval_trace.npz"), spec=spec.numpy(), keys=np.array(keys),
             mask_starts=np.array([masks[k][0][0] for k in keys]), mask_widths=np.array([masks[k][0][1] for k in keys]),
             logits=out, argmax=out.argmax(-1).astype(np.int32),
             param_checksum=np.array([float(sum(p.double().sum() for p in params)), float(sum(p.double().abs().sum() for p in params))]),
             model_seed=5, blank_bias=1.5, lr=1e-4)


def main():
    pc = reference_prepare_chunks()
    cases = []
    for spec_n, seq_len, overlap in [(360000, 16384, 14336), (415990, 16384, 14336), (16384, 16384, 14336),
                                     (1000, 16384, 14336), (1440000, 16384, 14336), (360000, 16384, 0), (360000, 2048, 0),
                                     (1500, 512, 256), (1200, 512, 256), (16385, 16384, 14336), (32768, 16384, 8192)]:
        data, keys = pc(torch.empty(1, 1, spec_n), seq_len, overlap)
        cases.append({"spec_n": spec_n, "seq_len": seq_len, "overlap": overlap, "n_windows": len(keys),
                      "first_keys": keys[:3], "last_key": keys[-1], "last_len": int(data[keys[-1]].shape[-1])})
    json.dump({"source": "reference lcasr/lib.py:128-145 executed via ast extraction", "cases": cases},
              open(os.path.join(HERE, "prepare_chunks.json"), "w"), indent=1)
    # hand-worked toy: seq_len 32, overlap 16, downsample 8 -> ds_len 4, overlap_ds 2; windows at 0, 16, 32 (last short: 24)
    json.dump({"seq_len": 32, "overlap": 16, "downsample": 8, "keys": [0, 16, 32], "u_lens": [32, 32, 24],
               "counts": [1, 1, 2, 2, 2, 2, 1]}, open(os.path.join(HERE, "stitch_toy.json"), "w"))
    shutil.copyfile(os.path.join(REF, "lcasr_nemo", "tokenizer.model"), os.path.join(HERE, "tokenizer_128.model"))
    # data asset: the 15 female + 15 male TEDLIUM talks of the reference's cross-gender evaluation
    shutil.copyfile(os.path.join(REF, "lcasr", "results", "gender_eval_tedlium", "speaker_manifest_15x15.json"),
                    os.path.join(HERE, "speaker_manifest_15x15.json"))
    from oracle.softdtw_ref import softdtw_forward_backward, sqdist
    torch.manual_seed(1234)
    a = torch.rand(4, 17, 2).numpy(); b = torch.rand(4, 15, 2).numpy()
    D = sqdist(a, b)
    val, grad = softdtw_forward_backward(D, 1.0, 0.0)
    np.savez(os.path.join(HERE, "softdtw_17x15x2.npz"), a=a, b=b, D=D, gamma=1.0, value=val, grad=grad)
    kernel_fixtures()
    print("golden fixtures written")


if __name__ == "__main__":
    main()
