"""Generates tests/golden/loop_pins.npz + loop_pins.json: outputs of the REFERENCE'S OWN LOOP FUNCTIONS, executed as a whole.

VERDICT r03 "missing" 2 / "next" 4: `oracle/dynamic_eval_ref.py`, `awmc_ref.py`, `wav2vec2_ref.py` are hand restatements; the step
ordering, the `/ (N * B)` scalings, the online / epochs interplay, `overlap_ds`, the label banks of AWMC, the clip -> step -> zero_grad
order of the wav2vec2 loop were pinned to nothing but the builder's reading.  Here the function DEFINITIONS are pulled out of the
reference files with `ast` and executed UNCHANGED:

  lcasr/lib.py:450-640                 dynamic_eval_ctc_loss      -> dyneval_*   (offline, online, 2 epochs, shuffle, return_params,
                                                                                  short recording, seq_len = -1 / overlap = -1 from the config)
  lcasr/lib.py:206-376                 AWMC                       -> awmc_*      (1 and 2 epochs per window)
  lcasr/run_half_concat_eval.py:64-160 adapt_on_concat_only       -> concat_*
  wav2vec2/lib.py:293-462              dynamic_eval_ctc_loss_su   -> su_*        (tiny Wav2Vec2ForCTC from `transformers`, the class the
                                                                                  reference loads)
  (+ the helpers they call that are plain torch / Python in the reference: prepare_chunks, the four get_*_from_args, frame_shuffle,
   add_random_noise, cutout, entropy_augmentation, disable_dropout — also executed unchanged.)

The reference's modules cannot be imported (un-vendored packages), so the LEAF names those functions take from them
are bound to the oracle's own restatements — this is what the pins do NOT cover:
  SpecAugment            -> stored masks derived from the window's content, applied by oracle.dynamic_eval_ref.apply_masks
  GreedyCTCDecoder       -> oracle.dynamic_eval_ref.greedy_ctc_ids + tokenizer.decode
  madgrad.MADGRAD        -> oracle.madgrad_ref.MADGRAD
  ExponentialMovingAverage (torch_ema) -> oracle.awmc_ref.EMARef
  the acoustic model     -> oracle.conformer_ref.SCConformerXLRef (toy size) / transformers.Wav2Vec2ForCTC (toy config)
  tqdm -> identity; augment.EffectChain / SoftDTW / plt: constructed or called by the _su loop but without effect on its outputs
  (the effect chains are built and never applied, wav2vec2/lib.py:391-412) -> inert objects.
What the pins DO cover: loop order and every piece of glue arithmetic between those leaves, as the reference wrote them.
`tests/test_reference_pins.py` (CPU) holds `oracle/*_ref.py` to these outputs; the HIP path is held to the oracle by the `-m gpu`
tests.  Run once in the build container: `python tests/golden/make_loop_pins.py` (the GPU box never sees /root/reference)."""
import argparse
import contextlib
import io
import json
import os
import platform
import random
import sys
import time
import types
from typing import Callable, Dict, List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_reference_pins import ref_functions  # noqa: E402
from oracle import dynamic_eval_ref as R  # noqa: E402
from oracle.awmc_ref import EMARef  # noqa: E402
from oracle.conformer_ref import SCConformerXLRef  # noqa: E402
from oracle.madgrad_ref import MADGRAD  # noqa: E402

from loop_pin_cases import *  # noqa: E402,F401,F403
from loop_pin_cases import (AWMC_CASES, CONCAT_CASES, CROSS_CASES, toy_records, DYNEVAL_CASES, SU_CASES, OracleGreedyCTCDecoder, StoredMaskSpecAugment, _Inert,
                            params_digest, quiet, tokenizer_128, toy_args, toy_model, VOCAB, TOY)  # noqa: E402


def lcasr_namespace():
    base = {"torch": torch, "nn": nn, "optim": optim, "F": F, "random": random, "time": time, "Callable": Callable, "Dict": Dict,
            "List": List, "tqdm": lambda it, **k: it, "madgrad": types.SimpleNamespace(MADGRAD=MADGRAD),
            "SpecAugment": StoredMaskSpecAugment, "GreedyCTCDecoder": OracleGreedyCTCDecoder, "ExponentialMovingAverage": EMARef}
    names = ["prepare_chunks", "get_specaugment_config_from_args", "get_frame_shuffle_config_from_args", "get_lr_args_from_args",
             "get_cutout_params_from_args", "frame_shuffle", "add_random_noise", "cutout", "entropy_augmentation",
             "dynamic_eval_ctc_loss", "AWMC"]
    return ref_functions("lcasr/lib.py", names, base)


def main():
    arrays, meta = {}, {"source": "reference loop functions executed unchanged via ast extraction (tests/golden/make_loop_pins.py); "
                                  "leaf classes bound to the oracle's restatements (see the generator's docstring)",
                        "machine": {"cpu": platform.processor() or platform.machine(), "torch": torch.__version__,
                                    "threads": torch.get_num_threads()}, "toy": TOY, "vocab": VOCAB}
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            meta["machine"]["cpu"] = line.split(":", 1)[1].strip()
            break
    ns = lcasr_namespace()
    tok = tokenizer_128()
    assert tok.vocab_size() == VOCAB

    # ---- dynamic_eval_ctc_loss (lcasr/lib.py:450-640)
    meta["dyneval"] = {}
    for tag, (T, seq_len, overlap, kw) in DYNEVAL_CASES.items():
        model = toy_model(seed=21)
        spec = torch.randn(1, 80, T, generator=torch.Generator().manual_seed(100 + T))
        before = [p.clone() for p in model.parameters()]
        random.seed(7); torch.manual_seed(9)
        out, params = quiet(ns["dynamic_eval_ctc_loss"], toy_args(**kw), model, spec, seq_len, overlap, tok, use_tqdm=False, return_params=True)
        assert all(torch.equal(a, b) for a, b in zip(before, model.parameters())), "the reference restores the weights (lib.py:636-637)"
        assert max((a - b).abs().max().item() for a, b in zip(before, params)) > 1e-6, "the adaptation must move the weights"
        arrays[f"dyneval_{tag}_out"] = out
        arrays[f"dyneval_{tag}_params"], psum = params_digest(params)
        meta["dyneval"][tag] = {"frames": T, "seq_len": seq_len, "overlap": overlap, "args": kw, "spec_seed": 100 + T, "model_seed": 21,
                                "random_seed": 7, "torch_seed": 9, "rows": int(out.shape[0]), "params_sum": psum}

    # ---- AWMC (lcasr/lib.py:206-376)
    meta["awmc"] = {}
    for tag, (T, seq_len, overlap, kw) in AWMC_CASES.items():
        model = toy_model(seed=22)
        spec = torch.randn(1, 80, T, generator=torch.Generator().manual_seed(200 + T))
        out, params = quiet(ns["AWMC"], toy_args(**kw), model, spec, seq_len, overlap, tok, use_tqdm=False, return_params=True)
        arrays[f"awmc_{tag}_out"] = out
        arrays[f"awmc_{tag}_params"], psum = params_digest(params)
        meta["awmc"][tag] = {"frames": T, "seq_len": seq_len, "overlap": overlap, "args": kw, "spec_seed": 200 + T, "model_seed": 22,
                             "rows": int(out.shape[0]), "params_sum": psum}

    # ---- adapt_on_concat_only (lcasr/run_half_concat_eval.py:64-160)
    lib_ns = types.SimpleNamespace(**{k: v for k, v in ns.items() if not k.startswith("__")})
    ns2 = ref_functions("lcasr/run_half_concat_eval.py", ["adapt_on_concat_only", "concatenate_specs"],
                        {"torch": torch, "random": random, "tqdm": lambda it, **k: it, "lib": lib_ns, "AWMC": ns["AWMC"],
                         "GreedyCTCDecoder": OracleGreedyCTCDecoder})
    meta["concat"] = {}
    for tag, lens, kw, adapt_overlap in CONCAT_CASES:
        model = toy_model(seed=23)
        g = torch.Generator().manual_seed(300 + len(lens))
        concat = ns2["concatenate_specs"]([torch.randn(1, 80, n, generator=g) for n in lens])
        params = quiet(ns2["adapt_on_concat_only"], toy_args(**kw), model, concat, tok, adapt_overlap=adapt_overlap)
        arrays[f"concat_{tag}_params"], psum = params_digest(params)
        meta["concat"][tag] = {"lens": list(lens), "args": kw, "adapt_overlap": adapt_overlap, "spec_seed": 300 + len(lens), "model_seed": 23,
                               "params_sum": psum}

    # ---- wav2vec2 dynamic_eval_ctc_loss_su (wav2vec2/lib.py:293-462)
    import tests_w2v2_toy as toy                                     # tokenizer / processor stand-ins shared with the CPU test
    base = {"torch": torch, "nn": nn, "optim": optim, "F": F, "random": random, "np": np, "tqdm": lambda it, **k: it,
            "madgrad": types.SimpleNamespace(MADGRAD=MADGRAD), "GreedyCTCDecoder": OracleGreedyCTCDecoder, "SoftDTW": _Inert(),
            "augment": _Inert(), "plt": _Inert()}
    ns3 = ref_functions("wav2vec2/lib.py", ["disable_dropout", "dynamic_eval_ctc_loss_su"], base)
    meta["su"] = {}
    for tag, kw, lr in SU_CASES:
        model = toy.model(seed=31)
        utts = toy.utterances(seed=41)
        random.seed(5)
        with contextlib.redirect_stdout(io.StringIO()):
            cwd = os.getcwd(); os.chdir("/tmp")                       # the loop writes loss.png through `plt` (inert here)
            try:
                out = ns3["dynamic_eval_ctc_loss_su"](argparse.Namespace(**kw), model, utts, 0, 0, toy.CharTokenizer(), toy.Processor(),
                                                      use_tqdm=False, optim=MADGRAD, lr_args={'lr': lr})
            finally:
                os.chdir(cwd)
        for k, u in enumerate(out):
            arrays[f"su_{tag}_probs{k}"] = u['probs'].numpy()
        meta["su"][tag] = {"args": kw, "lr": lr, "model_seed": 31, "utt_seed": 41, "random_seed": 5, "n": len(out)}

    # ---- the outer loop of BASELINE config 5 (lcasr/run_cross_dataset_eval.py:82-218: transcribe_from_logits, baseline_args, the repeat loop)
    from make_reference_pins import ref_statements
    from dynamic_asr_eval_amd.wer import basic_normalize
    from oracle.wer_ref import word_error_rate_detail
    code = [ref_statements("lcasr/run_cross_dataset_eval.py", 82, 94), ref_statements("lcasr/run_cross_dataset_eval.py", 96, 218)]
    meta["cross"] = {}
    for tag, lens_a, lens_b, kw in CROSS_CASES:
        model = toy_model(seed=24)
        args = toy_args(dataset="toy_a", dataset2="toy_b", **kw)
        scored = []

        def wer_recording(hypotheses, references):
            scored.append(list(hypotheses))
            return word_error_rate_detail(hypotheses=hypotheses, references=references)
        env = {"torch": torch, "argparse": argparse, "tqdm": lambda it, **k: it, "pickle": None, "args": args, "model": model,
               "data_a": toy_records(lens_a, 400), "data_b": toy_records(lens_b, 500), "eval_fn": ns["dynamic_eval_ctc_loss"], "tokenizer": tok,
               "beamsearch": None, "beams": 20, "decoder": OracleGreedyCTCDecoder(tok, VOCAB), "normalize": basic_normalize,
               "word_error_rate_detail": wer_recording, "original_model_params": [p.clone().detach().cpu() for p in model.parameters()],
               "adapt_overlap": kw["adapt_overlap"] if kw["adapt_overlap"] is not None else kw["overlap"]}
        with contextlib.redirect_stdout(io.StringIO()):
            for c in code:
                exec(c, env)
        r = env["results"]
        meta["cross"][tag] = {"lens_a": list(lens_a), "lens_b": list(lens_b), "args": kw, "model_seed": 24, "seeds": [400, 500],
                              "results": {k: r[k] for k in ("a_baseline", "b_baseline", "a_to_b", "a_to_a_loo")}, "scored_hypotheses": scored}

    np.savez_compressed(os.path.join(HERE, "loop_pins.npz"), **arrays)
    json.dump(meta, open(os.path.join(HERE, "loop_pins.json"), "w"), indent=1)
    print("loop pins written:", len(arrays), "arrays,", sum(a.nbytes for a in arrays.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
