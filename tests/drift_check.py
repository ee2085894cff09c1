#!/usr/bin/env python3
"""Multi-window parity drift at the BASELINE config-2 shape: N consecutive 16384-frame windows (overlap 14336)
of ONE recording through the weight-carrying adapt loop of reference lcasr/lib.py:537-581, SCConformerXL 6 x 768 / V+1 = 4096,
stored SpecAugment masks — HIP path (lib.dynamic_eval, online and offline) against the fp32 CPU oracle (oracle/dynamic_eval_ref.py)
and, as the noise-floor reference, the SAME oracle run in float64.

What the three-way comparison says: |hip - f64| next to |f32 oracle - f64| tells whether the HIP path is as close to the exact
arithmetic as the reference's own fp32 CPU arithmetic is; |hip - f32| is the number the 1e-3 bar of BASELINE.json is written for.
MADGRAD turns a gradient g into a step ~ lr^(2/3) * g^(1/3): its slope at g = 0 is unbounded, so the fp32 summation-order noise
of ANY two implementations is amplified where the true gradient is near zero (1.9e-6 on the parameters -> ~6e-4 on the log-probs
after ONE step at lr 9e-5).

Per 256-row band of the stitched output (one band = the 2048-frame stride): the online curve shows the drift after k adapt
steps (band b >= 8 is covered by windows b-7 ... only), the offline curve the final pass with the fully adapted weights.

  python tests/drift_check.py [--windows 8] [--fp64 1] [--grad_diag 1] [--spec_seed 77] [--mask_seed 9] [--out profiles/r04_drift_s77.json]

r04 — the full-length record of BASELINE config 2: `--windows 169 --fp64 0 --grad_diag 0 --budget_s 1000` runs ALL
169 windows of a 1 h recording (360 000 frames) through both sides; `--budget_s` first times the oracle on two windows and, if the
full recording would not fit the budget on this box's host cores, shortens the recording to the number of carried steps that does
(recorded as `windows`).  Besides the per-band curves the record holds the argmax mismatches with the ORACLE's own top-2 margin at
each of them and the edit distance between the two final transcripts (greedy CTC ids of the stitched output).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repository root
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

VOCAB, SEQ, OVL = 4095, 16384, 14336
BAND = (SEQ - OVL) // 8


def bands(a, b):
    """max |a - b| and argmax mismatches per band of BAND rows."""
    n = min(a.shape[0], b.shape[0])
    d = np.abs(a[:n].astype(np.float64) - b[:n].astype(np.float64)).max(-1)
    bad = a[:n].argmax(-1) != b[:n].argmax(-1)
    nb = (n + BAND - 1) // BAND
    return ([float(f"{d[k * BAND:(k + 1) * BAND].max():.3e}") for k in range(nb)], [int(bad[k * BAND:(k + 1) * BAND].sum()) for k in range(nb)])


def mismatch_margins(a, b):
    """Rows where the argmax differs, with the oracle's (b's) margin between its two largest log-probs there."""
    n = min(a.shape[0], b.shape[0])
    rows = np.nonzero(a[:n].argmax(-1) != b[:n].argmax(-1))[0]
    out = []
    for r in rows[:200]:
        top = np.sort(b[r])[-2:]
        out.append({"row": int(r), "oracle_margin": float(f"{top[1] - top[0]:.3e}"), "hip_id": int(a[r].argmax()), "oracle_id": int(b[r].argmax())})
    return out


def edit_distance(x, y):
    prev = list(range(len(y) + 1))
    for i, xi in enumerate(x, 1):
        cur = [i]
        for j, yj in enumerate(y, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (xi != yj)))
        prev = cur
    return prev[-1]


def greedy_ids(lp, blank):
    ids, prev = [], None
    for i in lp.argmax(-1).tolist():
        if i != blank and i != prev:
            ids.append(i)
        prev = i
    return ids


def with_progress(fn, timings, total, label):
    """Runs fn() in a worker thread and prints a progress line every 45 s (gpurun takes 7 silent minutes for a hang)."""
    import threading
    box = {}

    def run():
        try:
            box["out"] = fn()
        except BaseException as e:   # noqa: BLE001 — re-raised in the caller's thread
            box["err"] = e
    th = threading.Thread(target=run)
    th.start()
    t0 = time.time()
    while th.is_alive():
        th.join(45.0)
        if th.is_alive():
            print(f"[drift] {label}: {len(timings.get('adapt', []))}/{total} adapt steps, {len(timings.get('final', []))} final-pass windows, "
                  f"{time.time() - t0:.0f} s", flush=True)
    if "err" in box:
        raise box["err"]
    return box["out"]


class LabelLog:
    """Tokenizer wrapper that records what every tokenizer.encode() call returned: the pseudo-label ids each adapt step trained on.
    A near-tie that flips ONE frame's argmax between two fp32 implementations changes a window's target sequence, and from that step on
    the two runs follow different gradients — a divergence of another kind (1e-2) than rounding drift (1e-3)."""
    def __init__(self, tok):
        self.tok, self.log = tok, []

    def vocab_size(self):
        return self.tok.vocab_size()

    def decode(self, ids):
        return self.tok.decode(ids)

    def encode(self, text):
        ids = self.tok.encode(text)
        self.log.append(list(ids))
        return ids


def label_agreement(a, b):
    """Per adapt step: do both runs train on the same pseudo-label ids?  -> summary dict."""
    n = min(len(a), len(b))
    diff = [k for k in range(n) if a[k] != b[k]]
    return {"steps": n, "steps_with_different_labels": len(diff), "first_different_step": diff[0] if diff else None,
            "label_lengths": [len(x) for x in a[:n]][:200],
            "edit_distance_at_differing_steps": [edit_distance(a[k], b[k]) for k in diff[:50]]}


def make_args(masks, online, lr):
    ns = argparse.Namespace()
    ns.config = {'model': {'subsampling_factor': 8}, 'audio_chunking': {'size': SEQ, 'overlap': 0}, 'training': {'max_seq_len': 0}}
    ns.__dict__.update(dict(optim_lr=lr, epochs=1, shuffle=False, online=online, quiet=True, spec_augment_fixed_masks=masks, use_graphs=False))
    return ns


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=8)
    ap.add_argument("--fp64", type=int, default=1)
    ap.add_argument("--grad_diag", type=int, default=1)
    ap.add_argument("--lr", type=float, default=9e-5)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--spec_seed", type=int, default=77)
    ap.add_argument("--mask_seed", type=int, default=9)
    ap.add_argument("--blank_bias", type=float, default=1.34)
    ap.add_argument("--init", default="oracle", choices=["oracle", "synthetic"], help="synthetic: bench.py's weights (synthetic_weights.init_synthetic "
                    "seed 0 + the blank bias) instead of the oracle's seeded initialisation")
    ap.add_argument("--label_tokens", type=int, default=0, help="> 0: tokenizer.encode returns this many fixed seeded ids per window on BOTH sides "
                    "(bench.py's SubstituteLabelTokenizer): a seeded model's own labels collapse to blank after a few steps, this keeps every "
                    "carried step on a speech-like lattice (|alpha| in the thousands)")
    ap.add_argument("--budget_s", type=float, default=0.0, help="> 0: shorten the recording so that the fp32 oracle fits this many seconds")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_drift.json"))
    a = ap.parse_args()
    from oracle import dynamic_eval_ref as R
    from oracle.conformer_ref import SCConformerXLRef
    from oracle.madgrad_ref import MADGRAD as MADGRAD_REF
    from dynamic_asr_eval_amd import lib
    from dynamic_asr_eval_amd.datasets import synthetic_spec
    from dynamic_asr_eval_amd.model import SCConformerXL
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    torch.set_num_threads(a.threads)
    dev = torch.device("cuda:0")
    ref = SCConformerXLRef(vocab_size=VOCAB, seed=0, blank_bias=a.blank_bias)
    hip = SCConformerXL(vocab_size=VOCAB, device=dev)
    if a.init == "synthetic":
        from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
        init_synthetic(hip, seed=0, blank_bias=0.0)
        hip.P["decoder.ff.bias"][-1] += a.blank_bias
        ref.load_state_dict({k: v.cpu() for k, v in hip.state_dict().items()})
    else:
        hip.load_state_dict(ref.state_dict())
    tok = SyntheticTokenizer(VOCAB)

    def logged(log_tok):
        """With --label_tokens the loop trains on substituted ids; the log then records the model's OWN labels (the text hop it still makes)."""
        if a.label_tokens > 0:
            from bench import SubstituteLabelTokenizer
            return SubstituteLabelTokenizer(log_tok, a.label_tokens)
        return log_tok
    if a.label_tokens > 0:
        from bench import SubstituteLabelTokenizer
        tok = SubstituteLabelTokenizer(tok, a.label_tokens)
    res = {"what": __doc__.split("\n\n")[0], "lr": a.lr, "band_rows": BAND, "spec_seed": a.spec_seed, "mask_seed": a.mask_seed,
           "host_threads": a.threads, "requested_windows": a.windows,
           "label_tokens": a.label_tokens}
    if a.budget_s > 0:   # two oracle windows (adapt step + final-pass window each) give this box's per-window cost
        probe = synthetic_spec(SEQ + (SEQ - OVL), seed=a.spec_seed + 1)
        _, pk = R.prepare_chunks(probe, SEQ, OVL)
        gm = torch.Generator().manual_seed(1)
        tm = {}
        R.dynamic_eval_ref(ref, probe, SEQ, OVL, tok, MADGRAD_REF, {'lr': a.lr}, {}, fixed_masks={k: (R.draw_masks(6, 34, 80, gm), ([], [])) for k in pk},
                           max_windows=2, timings=tm)
        per = min(tm['adapt']) + min(tm['final'])
        fit = int(a.budget_s / (per * 1.05))
        res["oracle_probe_s_per_window"] = round(per, 2)
        print(f"[drift] oracle probe: {per:.2f} s per window (adapt step + final-pass window) -> {fit} windows fit {a.budget_s:.0f} s", flush=True)
        a.windows = max(2, min(a.windows, fit))
    # 169 windows = the 360 000-frame recording of BASELINE config 2 (prepare_chunks: 168 full windows + the 15 936-frame tail)
    n_frames = 360000 if a.windows >= 169 else SEQ + (a.windows - 1) * (SEQ - OVL)
    spec = synthetic_spec(n_frames, seed=a.spec_seed)
    _, keys = R.prepare_chunks(spec, SEQ, OVL)
    g = torch.Generator().manual_seed(a.mask_seed)
    masks = {k: (R.draw_masks(6, 34, 80, g), ([], [])) for k in keys}
    res.update({"windows": len(keys), "frames": n_frames, "keys": keys})

    t0 = time.time()
    log_hip, log_f32 = LabelLog(SyntheticTokenizer(VOCAB)), LabelLog(SyntheticTokenizer(VOCAB))
    hip_off = lib.dynamic_eval(make_args(masks, False, a.lr), hip, spec, SEQ, OVL, logged(log_hip), use_tqdm=False)
    hip_on = lib.dynamic_eval(make_args(masks, True, a.lr), hip, spec, SEQ, OVL, tok, use_tqdm=False)
    res["hip_seconds"] = round(time.time() - t0, 2)
    print(f"[drift] HIP done in {res['hip_seconds']} s", flush=True)
    t0 = time.time()
    tm32 = {}
    f32_off, f32_on, p32 = with_progress(lambda: R.dynamic_eval_ref(ref, spec, SEQ, OVL, logged(log_f32), MADGRAD_REF, {'lr': a.lr}, {}, fixed_masks=masks,
                                                                    also_online=True, return_params=True, timings=tm32), tm32, len(keys), "fp32 oracle")
    res["oracle_f32_seconds"] = round(time.time() - t0, 1)
    res["pseudo_labels_hip_vs_f32"] = label_agreement(log_hip.log, log_f32.log)
    print(f"[drift] fp32 oracle done in {res['oracle_f32_seconds']} s; pseudo-labels: {json.dumps({k: v for k, v in res['pseudo_labels_hip_vs_f32'].items() if k != 'label_lengths'})}", flush=True)
    for name, x, y in (("offline_hip_vs_f32", hip_off, f32_off), ("online_hip_vs_f32", hip_on, f32_on)):
        d, bad = bands(x, y)
        hyp, want = greedy_ids(x, VOCAB), greedy_ids(y, VOCAB)
        res[name] = {"max_abs_dlogp_per_band": d, "argmax_mismatch_per_band": bad, "max": max(d), "mismatches": sum(bad), "rows": int(min(x.shape[0], y.shape[0])),
                     "bands_over_1e-3": int(sum(v > 1e-3 for v in d)), "mismatch_rows": mismatch_margins(x, y),
                     "transcript": {"hip_tokens": len(hyp), "oracle_tokens": len(want), "edit_distance": edit_distance(hyp, want)}}
        print(f"[drift] {name}: max {max(d):.3e}, mismatches {sum(bad)}, transcript edits {res[name]['transcript']}; per band {d}", flush=True)
    if a.fp64:
        t0 = time.time()
        ref64 = SCConformerXLRef(vocab_size=VOCAB, seed=0, blank_bias=1.34).double()
        ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
        f64_off, f64_on = R.dynamic_eval_ref(ref64, spec.double(), SEQ, OVL, tok, MADGRAD_REF, {'lr': a.lr}, {}, fixed_masks=masks, also_online=True)
        res["oracle_f64_seconds"] = round(time.time() - t0, 1)
        for name, x, y in (("offline_hip_vs_f64", hip_off, f64_off), ("offline_f32_vs_f64", f32_off, f64_off),
                           ("online_hip_vs_f64", hip_on, f64_on), ("online_f32_vs_f64", f32_on, f64_on)):
            d, bad = bands(x, y)
            res[name] = {"max_abs_dlogp_per_band": d, "argmax_mismatch_per_band": bad, "max": max(d), "mismatches": sum(bad)}
            print(f"[drift] {name}: max {max(d):.3e}, mismatches {sum(bad)}; per band {d}", flush=True)
        del ref64

    if a.grad_diag:
        # one adapt step: which parameter's gradient is furthest from the float64 gradient, on either side
        from dynamic_asr_eval_amd import ops
        win = spec[:, :, :SEQ]
        chunk = win.repeat(2, 1, 1).clone()
        R.apply_masks(chunk[0], masks[0], zero_masking=False)

        def oracle_grads(model, x):
            model.zero_grad()
            out = model(audio_signal=x)['final_posteriors']
            ids = R.greedy_ctc_ids(out[-1].detach(), VOCAB)
            N = out.shape[1]
            loss = torch.nn.CTCLoss(blank=VOCAB, reduction='sum')(out[:1].transpose(0, 1), torch.LongTensor(ids)[None], torch.LongTensor([N]),
                                                                  torch.LongTensor([len(ids)])) / N
            loss.backward()
            return ids, {n: p.grad.detach().clone() for n, p in model.named_parameters()}, out.detach()

        ids32, g32, out32 = oracle_grads(ref, chunk)
        ref64 = SCConformerXLRef(vocab_size=VOCAB, seed=0, blank_bias=1.34).double()
        ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
        ids64, g64, out64 = oracle_grads(ref64, chunk.double())
        hip.use_graphs = False
        with torch.enable_grad():
            out = hip(audio_signal=chunk.to(dev))['final_posteriors']
        idd, nd = ops.ctc_greedy(out[-1].detach(), VOCAB)
        ids_h = idd[0, :int(nd[0])].cpu().tolist()
        N = out.shape[1]
        tg = torch.tensor([ids_h], dtype=torch.int32, device=dev)
        _, _, gl = ops.ctc_loss(out[:1].contiguous(), tg, torch.full((1,), N, dtype=torch.int32, device=dev),
                                torch.full((1,), len(ids_h), dtype=torch.int32, device=dev), VOCAB, reduction="sum", grad_scale=1.0 / N)
        hip.zero_grad()
        hip.backward(gl, n_active=1)
        rows = []
        for (n, _), gh in zip(hip.named_parameters(), hip.grads()):
            t = g64[n]
            den = t.abs().max().item() + 1e-300
            rows.append((n, (gh.cpu().double() - t).abs().max().item() / den, (g32[n].double() - t).abs().max().item() / den, den))
        rows.sort(key=lambda r: -r[1])
        res["grad_diag"] = {"labels_equal": ids_h == ids32 == ids64, "tokens": len(ids_h),
                            "fwd_max_abs_dlogp": {"hip_vs_f64": float((out.cpu().double() - out64).abs().max()), "f32_vs_f64": float((out32.double() - out64).abs().max())},
                            "worst_by_hip": [{"param": n, "hip_vs_f64_rel": float(f"{eh:.3e}"), "f32_vs_f64_rel": float(f"{e3:.3e}"), "max_abs_grad": float(f"{d:.3e}")}
                                             for n, eh, e3, d in rows[:12]],
                            "median_rel": {"hip_vs_f64": float(np.median([r[1] for r in rows])), "f32_vs_f64": float(np.median([r[2] for r in rows]))},
                            "max_rel": {"hip_vs_f64": max(r[1] for r in rows), "f32_vs_f64": max(r[2] for r in rows)}}
        print("[drift] grad diag:", json.dumps(res["grad_diag"], indent=1), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(f"[drift] wrote {a.out}")


if __name__ == "__main__":
    main()
