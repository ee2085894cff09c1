"""wav2vec2 harness (BASELINE config 3), shaped after the reference's `wav2vec2/tedlium/run.py:106-175`:
load model -> split the talk into utterances (`fetch_utterances`, :56-83) -> `lib.dynamic_eval_su` (:155) -> per-utterance
greedy decode + join (:156-160) -> WER (:169).

`AutoModelForCTC.from_pretrained("facebook/wav2vec2-base-960h")` and the TEDLIUM files cannot be fetched offline, so this
harness builds the base architecture (HF `Wav2Vec2Config()` defaults) with seeded weights, or loads a local HF
state_dict (`-c file.pt`, torch.load(weights_only=True)), and evaluates a synthetic TEDLIUM-shape talk: ~15 min cut into
utterances of 2-15 s.  Prints `WER:` like the reference and the audio-seconds per second."""
import argparse
import time

import torch

from . import wav2vec2_lib as lib
from .datasets import synthetic_waveform
from .decoding import GreedyCTCDecoder
from .wav2vec2_model import Wav2Vec2ForCTC
from .wer import word_error_rate_detail


def fetch_utterances_synthetic(total_seconds=900.0, seed=7, sample_rate=16000):
    """Utterance list like reference wav2vec2/tedlium/run.py:56-83 produces: dicts with 'waveform' [1, L] and 'text'."""
    g = torch.Generator().manual_seed(seed)
    utts, t = [], 0.0
    while t < total_seconds:
        d = float(2.0 + 13.0 * torch.rand(1, generator=g).item())
        d = min(d, total_seconds - t) if total_seconds - t > 2.0 else d
        wav = synthetic_waveform(d, seed=seed + len(utts), sample_rate=sample_rate)
        utts.append({'waveform': wav.unsqueeze(0), 'text': '', 'start': t, 'end': t + d})
        t += d
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


def main(args):
    device = torch.device('cuda', 0)
    model = Wav2Vec2ForCTC(None, device=device)
    if args.checkpoint:
        model.load_state_dict(torch.load(args.checkpoint, map_location='cpu', weights_only=True), strict=False)
    else:
        init_synthetic(model, args.seed)
    tokenizer = lib.CharTokenizer()
    utterances = fetch_utterances_synthetic(args.seconds, args.seed)
    a = argparse.Namespace(epochs=args.epochs, shuffle=False)
    torch.cuda.synchronize()
    t0 = time.time()
    utterances = lib.dynamic_eval_su(a, model, utterances, args.seq_len, args.overlap, tokenizer, None, use_tqdm=False,
                                     optim=lib.MADGRAD, lr_args={'lr': args.lr})
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=tokenizer.blank_id, device=device)
    text = " ".join(decoder(u['probs']) for u in utterances)
    torch.cuda.synchronize()
    dt = time.time() - t0
    wer = word_error_rate_detail([text.lower()], [" ".join(u['text'] for u in utterances).lower()])[0]
    print(f'WER: {wer}')
    print(f'wav2vec2 dynamic_eval_su: {len(utterances)} utterances, {args.seconds:.0f} s of audio in {dt:.2f} s -> {args.seconds / dt:.1f} audio-s/s')
    return wer


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('-c', '--checkpoint', default='')
    ap.add_argument('-seq', '--seq_len', type=int, default=131072)   # reference wav2vec2/lib.py:480-481 defaults
    ap.add_argument('-o', '--overlap', type=int, default=0)
    ap.add_argument('-epochs', '--epochs', type=int, default=1)
    ap.add_argument('--seconds', type=float, default=900.0)
    ap.add_argument('--lr', type=float, default=1e-6)
    ap.add_argument('--seed', type=int, default=0)
    main(ap.parse_args())
