"""wav2vec2 harness (BASELINE config 3) mirroring the reference's two wav2vec2 drivers, same flags (`lib.apply_args`,
reference wav2vec2/lib.py:477-493), same stdout lines (`Loaded model from`, `Total number of parameters`, `WER:`) and the same
`-log` line:
  --mode su       reference wav2vec2/tedlium/run.py:106-175: split the talk into STM utterances (`fetch_utterances`, :56-83) ->
                  `lib.dynamic_eval_su` (:155) -> per-utterance greedy decode, lower, strip, join (:156-160) -> normalise -> WER (:169)
  --mode chunked  reference wav2vec2/earnings22/run.py:59-117: `lib.dynamic_eval` over waveform windows (`-seq 131072 -overlap 0`,
                  :100) -> greedy decode of the stitched log-probs (:103) -> normalise -> WER (:114)
Like both reference drivers it processes the FIRST recording only (they `break` after it: tedlium/run.py:166, earnings22/run.py:111).

`AutoModelForCTC.from_pretrained("facebook/wav2vec2-base-960h")` and the corpora cannot be fetched offline, so the harness builds
the base architecture (HF `Wav2Vec2Config()` defaults: 7 conv layers of 512 channels, positional conv k=128 g=16, 12 x 768, vocab 32)
with seeded weights, or loads a local HF state_dict (`-c file.pt`, torch.load(weights_only=True)), and evaluates a synthetic
TEDLIUM-shape talk (`--seconds`, utterances of 2-15 s).  Reported WERs use this package's reduced text
normaliser (wer.basic_normalize), not whisper's EnglishTextNormalizer (un-vendored)."""
import argparse
import time

import torch

from . import wav2vec2_lib as lib
from .datasets import fetch_utterances_from_lines, synthetic_text, synthetic_waveform
from .decoding import GreedyCTCDecoder
from .wav2vec2_model import Wav2Vec2ForCTC
from .wer import basic_normalize as normalize, word_error_rate_detail


def synthetic_talk(total_seconds=900.0, seed=7, sample_rate=16000):
    """A TEDLIUM-shape talk: STM lines (utterances of 2-15 s, every 9th segment `ignore_time_segment_in_scoring`) and ONE waveform
    [1, L] for the whole talk, resident once; the utterances are cut out of it by the reference's own rule below."""
    g = torch.Generator().manual_seed(seed)
    lines, t, k = [], 0.0, 0
    while t < total_seconds:
        d = float(2.0 + 13.0 * torch.rand(1, generator=g).item())
        d = min(d, total_seconds - t) if total_seconds - t > 2.0 else d
        if k % 9 == 8:
            text = "ignore_time_segment_in_scoring"
        else:
            text = " ".join(w.replace("w", "word") for w in synthetic_text(max(1, int(2.5 * d)), seed + k).split())
        lines.append(f"talk{seed} 1 spk{seed} {t:.2f} {t + d:.2f} <o,f0,unknown> {text}")
        t, k = t + d, k + 1
    return lines, synthetic_waveform(t, seed=seed, sample_rate=sample_rate).unsqueeze(0)


def fetch_utterances_synthetic(total_seconds=900.0, seed=7, sample_rate=16000, stm_path=None):
    """Utterance list exactly as reference wav2vec2/tedlium/run.py:56-83 builds it (datasets.fetch_utterances, pinned to the
    reference's output): from `stm_path` when given (the audio itself cannot be decoded offline: a synthetic waveform of the STM's
    span stands in), else from a synthetic talk."""
    if stm_path:
        with open(stm_path, 'r') as f:
            lines = f.read().split('\n')
        ends = [float(l.split(' ')[4]) for l in lines if len(l.split(' ')) >= 6]
        wave = synthetic_waveform(max(ends) + 1.0, seed=seed, sample_rate=sample_rate).unsqueeze(0)
    else:
        lines, wave = synthetic_talk(total_seconds, seed, sample_rate)
    utts, _ = fetch_utterances_from_lines(lines, wave, sample_rate)
    return utts


def init_synthetic(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    for name, p in model.named_parameters():
        if p.dim() == 1:
            v = (1.0 if name.endswith("layer_norm.weight") or name.endswith("original0") else 0.0) + 0.02 * torch.randn(p.shape, generator=g)
        else:
            fan_in = p[0].numel()
            v = torch.randn(p.shape, generator=g) / fan_in ** 0.5
        p.copy_(v.to(p.device))


def replicate(model, n):
    """n - 1 more replicas of `model` (same configuration and weights) for lib.dynamic_eval_su_many."""
    out = [model]
    for _ in range(max(0, n - 1)):
        m = Wav2Vec2ForCTC(model.cfg, device=model.device)
        m.flat_params.copy_(model.flat_params)
        m.frozen = set(model.frozen)
        m.bucket_frames, m.graph_after, m.graph_budget_bytes = model.bucket_frames, model.graph_after, model.graph_budget_bytes
        out.append(m.eval())
    return out


def load_pretrained_model(args, device):
    """reference wav2vec2/lib.py:20-23 (`AutoModelForCTC.from_pretrained`) — offline: local state_dict or seeded weights."""
    model = Wav2Vec2ForCTC(None, device=device)
    if args.checkpoint:
        res = model.load_state_dict(torch.load(args.checkpoint, map_location='cpu', weights_only=True), strict=False)
        missing = getattr(res, 'missing_keys', [])
        if missing:
            raise KeyError(f'checkpoint {args.checkpoint}: {len(missing)} parameters of the model are missing (e.g. {missing[:3]})')
    else:
        init_synthetic(model, args.seed)
    return model, lib.CharTokenizer()


def main(args):
    assert args.split in ['test', 'dev'], f'Split must be either test or dev (got {args.split})'
    device = torch.device('cuda', 0)
    model, tokenizer = load_pretrained_model(args, device)
    print(f'Loaded model from {args.checkpoint}')
    print(f'Total number of parameters: {sum(p.numel() for p in model.parameters()) / 1e6:.2f}M')
    model.eval()
    tokenizer.blank_id = 0
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=tokenizer.blank_id, device=device)
    # the reference's driver walks the talks of the split, one dynamic_eval call each (tedlium/run.py:139-160); offline: `--talks` synthetic talks
    # (or the one `--stm` names).  Talks are independent (weights restored per call), so `--chains` of them can be in flight (lib.dynamic_eval_su_many)
    n_talks = 1 if args.stm else max(1, int(getattr(args, 'talks', 1)))
    talks = [fetch_utterances_synthetic(args.seconds, args.seed + (0 if args.split == 'test' else 1000) + 17 * t, stm_path=args.stm or None)
             for t in range(n_talks)]
    utterances = [u for talk in talks for u in talk]
    all_texts, all_golds = [], []
    torch.cuda.synchronize()
    t0 = time.time()
    chains = max(1, min(int(getattr(args, 'chains', 1)), n_talks))
    if args.mode == 'su' and chains > 1:
        outs = lib.dynamic_eval_su_many(args, replicate(model, chains), talks, args.seq_len, args.overlap, tokenizer, None, optim=lib.MADGRAD,
                                        lr_args={'lr': args.lr})
    for t, talk in enumerate(talks):
        gold_text = normalize(" ".join(u['text'] for u in talk)).lower()
        if args.mode == 'su':
            utterances_out = outs[t] if chains > 1 else lib.dynamic_eval_su(args, model, talk, args.seq_len, args.overlap, tokenizer, None, use_tqdm=False,
                                                                            optim=lib.MADGRAD, lr_args={'lr': args.lr})
            for i, utt in enumerate(utterances_out):
                utterances_out[i]['text'] = decoder(utt['probs']).lower()
            text = ' '.join([el['text'].strip() for el in utterances_out])
        else:
            audio_spec = torch.cat([u['waveform'] for u in talk], -1)                       # [1, L] waveform of the whole talk
            logits = lib.dynamic_eval(args, model, audio_spec, args.seq_len, args.overlap, tokenizer, None, use_tqdm=False,
                                      optim=lib.MADGRAD, lr_args={'lr': args.lr}, return_device=True)
            text = decoder(logits).lower()
        out = normalize(text).lower()
        if args.verbose:
            print(gold_text[:200], '\n', out[:200], '\n\n')
        all_texts.append(out)
        all_golds.append(gold_text)
    torch.cuda.synchronize()
    dt = time.time() - t0
    wer, words, ins_rate, del_rate, sub_rate = word_error_rate_detail(hypotheses=all_texts, references=all_golds)
    print(f'WER: {wer}')
    if args.log != '':
        with open(args.log, 'a') as f:
            f.write(f'{args.checkpoint}\t overlap: {args.overlap}\t seq_len: {args.seq_len}\t WER: {wer}\n')
    audio_s = sum(u['waveform'].shape[-1] for u in utterances) / 16000.0
    print(f'wav2vec2 {args.mode}: {len(utterances)} utterances, {audio_s:.0f} s of audio in {dt:.2f} s -> {audio_s / dt:.1f} audio-s/s')
    return wer


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', choices=['su', 'chunked'], default='su', help='su: tedlium/run.py (per utterance); chunked: earnings22/run.py')
    ap.add_argument('--seconds', type=float, default=900.0)
    ap.add_argument('--lr', type=float, default=1e-6)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--stm', type=str, default='', help='TEDLIUM .stm file: its segments define the utterances (reference tedlium/run.py:56-83)')
    ap.add_argument('--talks', type=int, default=1, help='synthetic talks of --seconds each (the reference walks the talks of the split)')
    ap.add_argument('--chains', type=int, default=1, help='talks in flight on the GPU (mode su): one model replica + HIP stream each')
    return ap


if __name__ == '__main__':
    main(lib.apply_args(build_parser()))
