"""Generates the committed fixtures under tests/golden/ (run once in the build container; the GPU box never sees
/root/reference).

  prepare_chunks.json   — key lists produced by the REFERENCE's own `prepare_chunks`: the function's source is pulled
                          out of /root/reference/lcasr/lib.py with `ast` (the module itself cannot be imported:
                          omegaconf/lcasr/... are absent) and executed as is; it is pure Python over tensor shapes.
  stitch_toy.json       — coverage counts of a 3-window toy worked by hand from reference lcasr/lib.py:615-629.
  tokenizer_128.model   — data asset copied from /root/reference/lcasr_nemo/tokenizer.model (SentencePiece, 128 pieces):
                          the only tokenizer the reference ships.
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
    from oracle.softdtw_ref import softdtw_forward_backward, sqdist
    torch.manual_seed(1234)
    a = torch.rand(4, 17, 2).numpy(); b = torch.rand(4, 15, 2).numpy()
    D = sqdist(a, b)
    val, grad = softdtw_forward_backward(D, 1.0, 0.0)
    np.savez(os.path.join(HERE, "softdtw_17x15x2.npz"), a=a, b=b, D=D, gamma=1.0, value=val, grad=grad)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
