"""Corpus-level word error rate with the return signature the reference harness unpacks:
`wer, words, ins_rate, del_rate, sub_rate = word_error_rate_detail(hypotheses=..., references=...)`
(reference lcasr/run_dynamic_eval_full.py:112-115; upstream `lcasr.eval.wer`, un-vendored).
Host-side integer dynamic programme (numpy rows); ties prefer substitution/match, then deletion, then insertion.
`edit_counts` exposes the four integer counters that the multi-GPU harness all-reduces over RCCL."""
import re

import numpy as np


def basic_normalize(text):
    """Minimal stand-in for whisper.normalizers.EnglishTextNormalizer (un-vendored, reference
    run_dynamic_eval_full.py:8,20,106): lower-case, drop punctuation, squeeze whitespace."""
    text = text.lower()
    text = re.sub(r"[^a-z0-9' ]+", " ", text)
    return re.sub(r"\s+", " ", text).strip()


def _align(hyp, ref):
    """(ins, del, sub) of the minimum-edit alignment of two token-id arrays."""
    n, m = len(ref), len(hyp)
    if n == 0:
        return m, 0, 0
    if m == 0:
        return 0, n, 0
    # cost and op-count rows over hypothesis positions; one row per reference word
    cost = np.arange(m + 1, dtype=np.int64)
    ins = np.arange(m + 1, dtype=np.int64)
    dele = np.zeros(m + 1, dtype=np.int64)
    sub = np.zeros(m + 1, dtype=np.int64)
    hyp = np.asarray(hyp)
    idx = np.arange(m + 1, dtype=np.int64)
    for i in range(1, n + 1):
        neq = (hyp != ref[i - 1]).astype(np.int64)
        diag = cost[:-1] + neq            # substitution / match from (i-1, j-1)
        up = cost[1:] + 1                 # deletion from (i-1, j)
        best = np.minimum(diag, up)
        from_diag = diag <= up
        b_ins = np.where(from_diag, ins[:-1], ins[1:])
        b_del = np.where(from_diag, dele[:-1], dele[1:] + 1)
        b_sub = np.where(from_diag, sub[:-1] + neq, sub[1:])
        # insertions chain along the row: n_cost[j] = min_{k<=j} (start[k] + j - k) with start[0] = i (column 0) and
        # start[k] = best[k-1]; a running minimum of start[k] - k plus the LAST k that attains it (fewest insertions
        # on ties) vectorises the scan.
        start = np.concatenate(([i], best))
        v = start - idx
        run = np.minimum.accumulate(v)
        k_star = np.maximum.accumulate(np.where(v == run, idx, 0))
        n_cost = run + idx
        extra = idx - k_star
        s_ins = np.concatenate(([0], b_ins)); s_del = np.concatenate(([i], b_del)); s_sub = np.concatenate(([0], b_sub))
        n_ins = s_ins[k_star] + extra
        n_del = s_del[k_star]
        n_sub = s_sub[k_star]
        cost, ins, dele, sub = n_cost, n_ins, n_del, n_sub
    return int(ins[m]), int(dele[m]), int(sub[m])


def edit_counts(hypotheses, references):
    """(insertions, deletions, substitutions, reference_words) summed over the corpus."""
    tot = [0, 0, 0, 0]
    for h, r in zip(hypotheses, references):
        hw, rw = h.split(), r.split()
        vocab = {}
        hi = [vocab.setdefault(w, len(vocab)) for w in hw]
        ri = [vocab.setdefault(w, len(vocab)) for w in rw]
        i, d, s = _align(hi, ri)
        tot[0] += i; tot[1] += d; tot[2] += s; tot[3] += len(rw)
    return tuple(tot)


def rates_from_counts(ins, dele, sub, words):
    if words == 0:
        return (float("inf") if (ins + dele + sub) else 0.0), 0, 0.0, 0.0, 0.0
    return (ins + dele + sub) / words, words, ins / words, dele / words, sub / words


def word_error_rate_detail(hypotheses, references):
    return rates_from_counts(*edit_counts(hypotheses, references))
