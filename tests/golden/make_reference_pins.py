"""Generates tests/golden/reference_pins.npz + reference_pins.json: outputs of the REFERENCE'S OWN code for every piece of
the dynamic-eval path that is pure torch / Python in the reference (no un-vendored dependency), so the oracle restatement
and the HIP kernels are held to numbers the reference itself produced.  Run once in the build container
(`python tests/golden/make_reference_pins.py`); the GPU box never sees /root/reference.

How: the reference modules cannot be imported (`import lib` -> ModuleNotFoundError: omegaconf, lcasr, ...), so the
function definitions / statement ranges are pulled out of the reference files with `ast` and executed AS THEY ARE, with
nothing but `torch`, `re`, `random` and `typing.List` in their namespace — no stand-ins for absent packages.

  lcasr/lib.py:81-84      frame_shuffle                       -> frame_shuffle_{t,f,tf}
  lcasr/lib.py:379-382    add_random_noise                    -> noise_0p3
  lcasr/lib.py:384-417    cutout (mean / mean_recording / zero)-> cutout_*
  lcasr/lib.py:102-125    get_specaugment/frame_shuffle/lr_*_from_args, :419-428 get_cutout_params_from_args -> json
  lcasr/lib.py:615-629    the stitch statements of dynamic_eval_ctc_loss, run on a seeded `model_outputs` dict with a short
                          tail window                         -> stitch_inner_*
  lcasr/run_seq_eval.py:130-144  the outer stitch statements  -> stitch_outer_*
  lcasr/tedlium/run.py:25-51     open_stm + proc_stm_and_timings on a synthetic STM file -> json
  wav2vec2/tedlium/run.py:25-83  open_stm + fetch_utterances on the same STM lines and a ramp waveform -> json (fetch_utterances)
  wav2vec2/soft_dtw_cuda.py:319-329  SoftDTW._euclidean_dist_func -> sqdist_*
  lcasr/enc_dec_teacher_filters.py   add_enc_dec_teacher_filter_args defaults + should_skip_faulty_teacher_prediction decisions -> json
"""
import argparse
import ast
import json
import os
import random
import re
import sys
import tempfile
from typing import List

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def _tree(rel):
    return ast.parse(open(os.path.join(REF, rel)).read())


def ref_functions(rel, names, ns=None):
    """Top-level (or class-level) function definitions of a reference file, compiled unchanged."""
    ns = dict(ns or {})
    found = {}
    for node in ast.walk(_tree(rel)):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
            node.decorator_list = []          # @staticmethod of a method extracted as a plain function
            found[node.name] = node
    missing = set(names) - set(found)
    assert not missing, f"{rel}: {missing} not found"
    body = [found[n] for n in names]
    exec(compile(ast.Module(body=body, type_ignores=[]), f"reference:{rel}", "exec"), ns)
    return ns


def ref_statements(rel, first, last):
    """The statements of a reference file that lie wholly inside [first, last] (outermost ones), compiled unchanged."""
    out = []

    def visit(node):
        for child in ast.iter_child_nodes(node):
            if isinstance(child, ast.stmt) and child.lineno >= first and child.end_lineno <= last:
                out.append(child)
            else:
                visit(child)

    visit(_tree(rel))
    out.sort(key=lambda n: n.lineno)
    assert out and out[0].lineno == first and out[-1].end_lineno == last, (rel, first, last, [(n.lineno, n.end_lineno) for n in out])
    return compile(ast.Module(body=out, type_ignores=[]), f"reference:{rel}:{first}-{last}", "exec")


def stitch_case(seed, n_windows, ds_full, ds_tail, overlap_ds, C, spec_n_quarter_plus):
    """A seeded `model_outputs` dict as the loops build it (lib.py:583-589,604-609): probabilities per window, a short tail."""
    g = torch.Generator().manual_seed(seed)
    keys = [k * 100 for k in range(n_windows)]
    mo = {}
    for j, k in enumerate(keys):
        ds = ds_tail if j == n_windows - 1 else ds_full
        mo[k] = {'logits': torch.softmax(torch.randn(1, ds, C, generator=g), -1), 'ds_len': ds, 'overlap_ds': overlap_ds}
    return keys, mo, torch.zeros(1, spec_n_quarter_plus, C), torch.zeros(1, spec_n_quarter_plus, C)


def main():
    arrays, meta = {}, {"source": "reference functions / statements executed unchanged via ast extraction (tests/golden/make_reference_pins.py)"}
    base = {"torch": torch, "random": random, "re": re, "List": List}

    # ---- augmentations (lcasr/lib.py:81-84, 379-417)
    ns = ref_functions("lcasr/lib.py", ["frame_shuffle", "add_random_noise", "cutout"], base)
    spec = torch.randn(1, 80, 600, generator=torch.Generator().manual_seed(80)) * 1.7 + 0.2
    arrays["aug_spec"] = spec.numpy()
    for tag, kw in (("t", dict(time_dimension=True)), ("f", dict(freq_dimension=True)), ("tf", dict(time_dimension=True, freq_dimension=True))):
        torch.manual_seed(5)
        arrays[f"frame_shuffle_{tag}"] = ns["frame_shuffle"](spec.clone(), **kw).numpy()
    torch.manual_seed(6)
    arrays["noise_0p3"] = ns["add_random_noise"](spec.clone(), 0.3).numpy()
    for val in ("mean", "mean_recording", "zero"):
        torch.manual_seed(7)
        arrays[f"cutout_{val}"] = ns["cutout"](spec.clone(), 600, cutout_val=val, num_rectangles=7, max_width=100, max_height=10).numpy()
    torch.manual_seed(8)   # shorter than the tuning length: the rectangle count scales down (lib.py:393)
    arrays["cutout_short"] = ns["cutout"](spec[:, :, :250].clone(), 600, cutout_val="mean", num_rectangles=7, max_width=40, max_height=6).numpy()
    meta["aug_seeds"] = {"frame_shuffle": 5, "noise": 6, "cutout": 7, "cutout_short": 8}

    # ---- arg -> config helpers (lcasr/lib.py:102-125, 419-428)
    ns = ref_functions("lcasr/lib.py", ["get_specaugment_config_from_args", "get_frame_shuffle_config_from_args", "get_lr_args_from_args",
                                        "get_cutout_params_from_args"], base)
    arg_cases = [
        {},
        {"spec_augment_n_freq_masks": 6, "spec_augment_freq_mask_param": 34, "spec_augment_n_time_masks": 0, "optim_lr": 9e-6},
        {"spec_augment_zero_masking": True, "spec_augment_min_p": 0.1, "spec_augment_time_mask_param": 20, "spec_augment_n_time_masks": 2,
         "frame_shuffle_time_dimension": True, "optim_lr": 1e-4, "optim_weight_decay": 0.01, "cutout_value": "zero", "cutout_num_rectangles": 9,
         "cutout_max_width": 50, "cutout_max_height": 4, "epochs": 3},
    ]
    meta["arg_cases"] = []
    for case in arg_cases:
        a = argparse.Namespace(**case)
        meta["arg_cases"].append({"args": case, "specaugment": ns["get_specaugment_config_from_args"](a),
                                  "frame_shuffle": ns["get_frame_shuffle_config_from_args"](a), "lr": ns["get_lr_args_from_args"](a),
                                  "cutout": ns["get_cutout_params_from_args"](a, 16384)})

    # ---- stitch statements: inner (lcasr/lib.py:615-629) and outer (lcasr/run_seq_eval.py:130-144)
    for tag, rel, first, last in (("inner", "lcasr/lib.py", 615, 629), ("outer", "lcasr/run_seq_eval.py", 130, 144)):
        code = ref_statements(rel, first, last)
        cases = []
        for ci, (nw, ds_full, ds_tail, ov, C, rows) in enumerate([(4, 64, 41, 48, 17, 400), (3, 32, 32, 0, 9, 200), (1, 50, 50, 30, 5, 120),
                                                                  (5, 40, 7, 35, 6, 300)]):
            keys, mo, acc, cnt = stitch_case(100 + ci, nw, ds_full, ds_tail, ov, C, rows)
            env = dict(base, model_outputs=mo, all_logits=acc, logit_count=cnt)
            exec(code, env)
            out = env["logits"]
            assert out.dim() == 3 and out.shape[0] == 1
            arrays[f"stitch_{tag}_{ci}_in"] = np.concatenate([mo[k]['logits'][0].numpy() for k in keys], 0)
            arrays[f"stitch_{tag}_{ci}_out"] = out[0].numpy()
            cases.append({"keys": keys, "ds_len": [mo[k]['ds_len'] for k in keys], "overlap_ds": ov, "classes": C, "acc_rows": rows,
                          "out_rows": int(out.shape[1])})
        meta[f"stitch_{tag}"] = cases

    # ---- TEDLIUM STM text handling (lcasr/tedlium/run.py:25-51)
    ns = ref_functions("lcasr/tedlium/run.py", ["open_stm", "proc_stm_and_timings"], base)
    stm_lines = [
        "talk1 1 spk1 0.00 4.25 <o,f0,male> hello there it 's me",
        "talk1 1 spk1 4.25 9.00 <o,f0,male> ignore_time_segment_in_scoring",
        "talk1 1 spk1 9.00 15.50 <o,f0,male> we  did n't  know  what 's next",
        "short line",
        "",
        "talk1 1 spk1 15.50 21.75 <o,f0,male> the speaker 's point",
        "talk1 1 spk1 21.75 30.00 <o,f0,male> ignore_time_segment_in_scoring",
        "talk1 1 spk1 30.00 33.10 <o,f0,male> thank you",
    ]
    with tempfile.NamedTemporaryFile("w", suffix=".stm", delete=False) as f:
        f.write("\n".join(stm_lines))
        path = f.name
    text, timings, remove = ns["proc_stm_and_timings"](path)
    os.unlink(path)
    meta["stm"] = {"lines": stm_lines, "text": text, "timings": timings, "remove_timings": remove}

    # ---- wav2vec2 per-utterance slicing (wav2vec2/tedlium/run.py:56-83): STM lines -> utterance dicts with waveform views
    ns = ref_functions("wav2vec2/tedlium/run.py", ["open_stm", "fetch_utterances"], base)
    sr = 16000
    wave = torch.arange(34 * sr, dtype=torch.float32)[None] * 0.5            # a ramp: every slice is identified by its first / last value
    with tempfile.NamedTemporaryFile("w", suffix=".stm", delete=False) as f:
        f.write("\n".join(stm_lines))
        path = f.name
    utts, all_text = ns["fetch_utterances"](path, wave, sr)
    os.unlink(path)
    meta["fetch_utterances"] = {"sample_rate": sr, "wave_samples": int(wave.shape[1]), "wave_rule": "0.5 * arange", "all_text": all_text,
                                "utterances": [{"start": u["start"], "end": u["end"], "text": u["text"], "start_frame": u["start_frame"],
                                                "end_frame": u["end_frame"], "shape": list(u["waveform"].shape),
                                                "first": float(u["waveform"][0, 0]), "last": float(u["waveform"][0, -1])} for u in utts]}

    # ---- squared Euclidean distance of the soft-DTW module (wav2vec2/soft_dtw_cuda.py:319-329)
    ns = ref_functions("wav2vec2/soft_dtw_cuda.py", ["_euclidean_dist_func"], base)
    g = torch.Generator().manual_seed(31)
    for tag, (B, N, M, D) in (("a", (2, 17, 15, 2)), ("b", (1, 40, 33, 32)), ("c", (3, 5, 9, 7))):
        x, y = torch.randn(B, N, D, generator=g), torch.randn(B, M, D, generator=g)
        arrays[f"sqdist_{tag}_x"], arrays[f"sqdist_{tag}_y"] = x.numpy(), y.numpy()
        arrays[f"sqdist_{tag}_d"] = ns["_euclidean_dist_func"](x, y).numpy()

    # ---- enc-dec teacher filters (lcasr/enc_dec_teacher_filters.py): flag defaults and decisions on a table of cases
    from difflib import SequenceMatcher
    ns = ref_functions("lcasr/enc_dec_teacher_filters.py",
                       ["add_enc_dec_teacher_filter_args", "_sequence_similarity", "_word_sequence", "_longest_consecutive_repeat",
                        "_find_repeated_ngram_loop", "should_skip_faulty_teacher_prediction"], dict(base, SequenceMatcher=SequenceMatcher))
    defaults = vars(ns["add_enc_dec_teacher_filter_args"](argparse.ArgumentParser()).parse_args([]))
    meta["teacher_filter_defaults"] = defaults
    cases = []
    table = [
        (dict(teacher_filter_max_length=True), list(range(40)), "a b c", 256, {}),
        (dict(teacher_filter_max_length=True), list(range(20)), "a b c", 256, {}),
        (dict(teacher_filter_max_length=True, teacher_min_frames_per_token=4), list(range(40)), "a b c", 256, {}),
        (dict(teacher_filter_max_consecutive_token_repeat=True), [5, 7, 7, 7, 7, 2], "x", 512, {}),
        (dict(teacher_filter_max_consecutive_token_repeat=True), [5, 7, 7, 7, 2], "x", 512, {}),
        (dict(teacher_filter_max_consecutive_token_repeat=True, teacher_max_consecutive_token_repeat=1), [1, 2, 2, 3], "x", 512, {}),
        (dict(teacher_filter_repeated_token_ngrams=True), [1, 2, 1, 2, 9], "x", 512, {}),
        (dict(teacher_filter_repeated_token_ngrams=True), [1, 2, 3, 1, 2, 4], "x", 512, {}),
        (dict(teacher_filter_repeated_token_ngrams=True, teacher_repeated_token_ngram_sizes=[3], teacher_repeated_token_ngram_min_repeats=3),
         [4, 5, 6, 4, 5, 6, 4, 5, 6, 1], "x", 512, {}),
        (dict(teacher_filter_low_confidence=True), [1, 2], "x", 512, dict(teacher_mean_max_prob=0.2, teacher_mean_entropy=1.0)),
        (dict(teacher_filter_low_confidence=True), [1, 2], "x", 512, dict(teacher_mean_max_prob=0.9, teacher_mean_entropy=3.1)),
        (dict(teacher_filter_low_confidence=True), [1, 2], "x", 512, dict(teacher_mean_max_prob=0.9, teacher_mean_entropy=0.4)),
        (dict(teacher_filter_repeated_words=True), [1], "the the the the cat", 512, {}),
        (dict(teacher_filter_repeated_words=True), [1], "the the the cat", 512, {}),
        (dict(teacher_filter_ctc_agreement=True), [1], "hello big world again", 512, dict(ctc_text="hello world")),
        (dict(teacher_filter_ctc_agreement=True), [1], "hello big world again", 512, dict(ctc_text="completely different words here now")),
        (dict(), list(range(500)), "the the the the the", 16, dict(teacher_mean_max_prob=0.0)),
        (dict(teacher_filter_max_length=True, teacher_filter_repeated_words=True), [], "", 100, {}),
    ]
    for flags, tokens, text, frames, extra in table:
        a = argparse.Namespace(**dict(defaults, **flags))
        skip, reason = ns["should_skip_faulty_teacher_prediction"](args=a, teacher_pred_tokens=tokens, teacher_pred_text=text, spec_frames=frames, **extra)
        cases.append({"flags": flags, "tokens": tokens, "text": text, "frames": frames, "extra": extra, "skip": bool(skip), "reason": reason})
    meta["teacher_filter_cases"] = cases

    np.savez_compressed(os.path.join(HERE, "reference_pins.npz"), **arrays)
    json.dump(meta, open(os.path.join(HERE, "reference_pins.json"), "w"), indent=1)
    print("reference pins written:", len(arrays), "arrays")


if __name__ == "__main__":
    main()
