"""ORACLE (test infrastructure only).  Corpus-level word error rate with the return signature the reference harnesses unpack —
`wer, words, ins_rate, del_rate, sub_rate = word_error_rate_detail(hypotheses=..., references=...)` (reference
lcasr/run_dynamic_eval_full.py:112-115, run_cross_dataset_eval.py:118,138,176,194).  The function itself lives in the un-vendored
`lcasr.eval.wer` (absent: PARITY UNPINNED); restated as the textbook minimum-edit alignment, plain Python loops, one table per
utterance pair, ties resolved match/substitution before deletion before insertion (the product's numpy rows use the same order;
tests/test_host_cpu.py compares the two on random corpora)."""


def _align(hyp, ref):
    n, m = len(ref), len(hyp)
    # cell = (cost, ins, del, sub)
    prev = [(j, j, 0, 0) for j in range(m + 1)]
    for i in range(1, n + 1):
        cur = [(i, 0, i, 0)]
        for j in range(1, m + 1):
            neq = int(hyp[j - 1] != ref[i - 1])
            d = prev[j - 1]; u = prev[j]; l = cur[j - 1]
            best = (d[0] + neq, d[1], d[2], d[3] + neq)
            if u[0] + 1 < best[0]:
                best = (u[0] + 1, u[1], u[2] + 1, u[3])
            if l[0] + 1 < best[0]:
                best = (l[0] + 1, l[1] + 1, l[2], l[3])
            cur.append(best)
        prev = cur
    return prev[m][1], prev[m][2], prev[m][3]


def edit_counts(hypotheses, references):
    tot = [0, 0, 0, 0]
    for h, r in zip(hypotheses, references):
        hw, rw = h.split(), r.split()
        i, d, s = _align(hw, rw)
        tot[0] += i; tot[1] += d; tot[2] += s; tot[3] += len(rw)
    return tuple(tot)


def word_error_rate_detail(hypotheses, references):
    ins, dele, sub, words = edit_counts(hypotheses, references)
    if words == 0:
        return (float("inf") if (ins + dele + sub) else 0.0), 0, 0.0, 0.0, 0.0
    return (ins + dele + sub) / words, words, ins / words, dele / words, sub / words
