"""Pieces shared by the harness mirrors (run_dynamic_eval_full / run_cross_dataset_eval / run_whole_concat_eval)."""
import torch

from . import dist as ddist
from .wer import basic_normalize, edit_counts, rates_from_counts

normalize = basic_normalize


def set_params(model, params):
    """`for p, u in zip(model.parameters(), updated): p.data = u.data.to(p.device)` of the reference harnesses
    (run_cross_dataset_eval.py:157-158,197-198; run_whole_concat_eval.py:149-150), kept inside the flat HBM buffer."""
    for p, u in zip(model.parameters(), params):
        p.copy_(u.to(p.device))


def clone_params(model):
    return model.flat_params.clone() if hasattr(model, "flat_params") else [p.clone().detach() for p in model.parameters()]


def restore_params(model, snap):
    if hasattr(model, "flat_params"):
        model.flat_params.copy_(snap)
    else:
        set_params(model, snap)


def score_texts(preds, golds, reduce_over_ranks=False):
    counts = edit_counts(preds, golds)
    if reduce_over_ranks:
        counts = ddist.all_reduce_counts(counts)       # RCCL: 4 int64 counters
    wer, words, ins_rate, del_rate, sub_rate = rates_from_counts(*counts)
    return {"wer": wer, "words": words, "ins_rate": ins_rate, "del_rate": del_rate, "sub_rate": sub_rate}


def transcribe(decoder, logits):
    return normalize(decoder(logits)).lower()
