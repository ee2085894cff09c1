"""Drop-in mirror of the reference's wav2vec2 dynamic-eval library (reference wav2vec2/lib.py) on the HIP path:
`dynamic_eval_ctc_loss_su` / `dynamic_eval_su` (:293-462, the loop `wav2vec2/tedlium/run.py:155` drives) and the chunked
`dynamic_eval_ctc_loss` / `dynamic_eval` (:41-235), plus `disable_dropout` (:34-38).

Same call signatures; `model` is this package's Wav2Vec2ForCTC (wav2vec2_model.py), `processor` may be None: the only
thing the reference uses it for is `processor.feature_extractor(...)` = zero-mean / unit-variance normalisation of each
waveform (eps 1e-7), which runs here as a HIP kernel on the device (dyn_colnorm_fwd with one channel).
The WavAugment chain the chunked variant applies to its first copies (`augment.EffectChain`, lib.py:144-156; the package is un-vendored
and absent) is restated from the published effects as far as it needs no external tool: 100 x `time_dropout(0.1 s)` and the zero-noise
`additive_noise(snr=0)` run on the device (`wav_augment_chunk`; `-kwargs`-style switch `args.wav_augment=False` turns it off); the
chain's last effect, `.reverb(50, 50, 100)`, is a sox effect and is not reproduced (parity unpinned).  The per-utterance variant builds
its effect chains without applying them (lib.py:391-412), so its copies are clean in the reference too."""
import random
from types import SimpleNamespace

import torch

from . import ops
from .decoding import GreedyCTCDecoder
from .optim import MADGRAD, Adam  # noqa: F401

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, **_):
        return x


class CharTokenizer:
    """Stand-in for the HF Wav2Vec2CTCTokenizer of wav2vec2-base-960h (needs downloaded files): same 32-symbol layout
    (<pad>=0 is the CTC blank, <s>, </s>, <unk>, '|' word delimiter, then letters and apostrophe), `blank_id`,
    `vocab_size`, `decode(ids) -> str`, `tokenizer(text).input_ids`."""
    SYMBOLS = ["<pad>", "<s>", "</s>", "<unk>", "|"] + list("ETAONIHSRDLUMWCFGYPBVK'XJQZ")

    def __init__(self):
        self.vocab = {s: i for i, s in enumerate(self.SYMBOLS)}
        self.blank_id = 0
        self.vocab_size = len(self.SYMBOLS)

    def decode(self, ids):
        return "".join(" " if self.SYMBOLS[i] == "|" else self.SYMBOLS[i] for i in ids if i > 3 or i == 3)

    def __call__(self, text):
        ids = [self.vocab["|"] if ch == " " else self.vocab.get(ch, 3) for ch in text if ch == " " or ch in self.vocab or True]
        return SimpleNamespace(input_ids=ids)


def disable_dropout(model):
    """reference wav2vec2/lib.py:34-38 — our model has no dropout modules (eval-mode forward)."""
    return model


def normalize_waveform(x):
    """processor.feature_extractor(...) of the reference (lib.py:161,406) on the device: per-row zero mean / unit variance."""
    B, L = x.shape
    ones = torch.ones(1, device=x.device, dtype=torch.float32)
    zeros = torch.zeros(1, device=x.device, dtype=torch.float32)
    y, _, _ = ops.colnorm(x.contiguous().view(B, L, 1), ones, zeros, eps=1e-7)
    return y.view(B, L)


def time_dropout_draws(n_samples, n_rounds=100, max_seconds=0.1, rate=16000):
    """The (start, length) pairs of `n_rounds` applications of WavAugment's `EffectChain().time_dropout(max_seconds)` to a waveform of
    `n_samples` samples (reference wav2vec2/lib.py:145,154-155: `for _ in range(100): augmentation_1.apply(...)`).  `augment`
    (facebookresearch/WavAugment) is un-vendored; its TimeDropout is restated from the published effect: `length = np.random.randint(0,
    max_frames)`, `start = np.random.randint(0, max(1, n - length))`, the span is zeroed; max_frames = int(rate * max_seconds).
    PARITY UNPINNED against the upstream package (no fixture); host numpy RNG as upstream, so `np.random.seed` reproduces a run."""
    import numpy as np
    max_frames = int(rate * max_seconds)
    starts, lengths = [], []
    for _ in range(n_rounds):
        length = int(np.random.randint(0, max_frames))
        start = int(np.random.randint(0, max(1, n_samples - length)))
        starts.append(start); lengths.append(length)
    return starts, lengths


def wav_augment_chunk(wave_row):
    """The reference's WavAugment chain on one copy [L] of a waveform window, in place on the device (wav2vec2/lib.py:144-156): 100 x
    time_dropout(0.1 s), then `additive_noise(zeros, snr=0)` = 0.5 * x (WavAugment mixes r / (1 + r) * x + 1 / (1 + r) * noise with r =
    10^(snr / 10) = 1; the scale survives the feature extractor's normalisation only through its 1e-7 epsilon).  NOT applied: `.reverb(50,
    50, 100)` — a sox effect; neither WavAugment nor sox exists here and the reference holds no output of it (parity unpinned, stated)."""
    starts, lengths = time_dropout_draws(wave_row.numel())
    t0 = torch.tensor(starts, dtype=torch.int32, device=wave_row.device)
    wd = torch.tensor(lengths, dtype=torch.int32, device=wave_row.device)
    ops.specaug_timemask(wave_row.view(1, -1), t0, wd, 0.0)
    ops.axpby(wave_row, wave_row, a=0.0, b=0.5)
    return wave_row


def _snapshot(model):
    return model.flat_params.clone()


def dynamic_eval_ctc_loss_su(args, model, utterances, seq_len, overlap, tokenizer, processor, use_tqdm=True, optim=MADGRAD,
                             num_negatives=1, lr_args={'lr': 1e-15}, ngram_decoder=None):
    """reference wav2vec2/lib.py:293-462"""
    if ngram_decoder is not None:
        raise NotImplementedError("n-gram (pyctcdecode) pseudo-labels need the un-vendored decoder and its ARPA file")
    device = model.device
    downsampling_factor = 4
    original = _snapshot(model)
    blank = tokenizer.blank_id
    optimizer = optim(model.parameters(), **lr_args)
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=blank, device=device)
    assert overlap / downsampling_factor == overlap // downsampling_factor, 'Overlap must be a multiple of the downsampling factor'
    model = disable_dropout(model)
    # hipGraph replay over utterance-length buckets (wav2vec2_model.py::forward): `args.use_graphs` (default on), `args.bucket_frames`
    was_graphs = model.use_graphs
    model.use_graphs = bool(args.__dict__.get('use_graphs', True))
    model.bucket_frames = int(args.__dict__.get('bucket_frames', model.bucket_frames))
    for epoch in range(args.__dict__.get('epochs', 1)):
        indexes = list(range(len(utterances)))
        indexes = random.sample(indexes, len(indexes)) if args.__dict__.get('shuffle', False) else indexes
        pbar = tqdm(indexes) if use_tqdm else indexes
        for idx in pbar:
            wav = utterances[idx]['waveform'].to(device=device, dtype=torch.float32)            # [1, L]
            audio_chunk = wav.reshape(1, -1).repeat(num_negatives + 1, 1).contiguous()          # [B, L]
            input_values = normalize_waveform(audio_chunk)
            with torch.enable_grad():
                out = model(input_values)
            log_p = ops.log_softmax(out.logits)                                                  # F.log_softmax, lib.py:417
            N = out.frames                      # < log_p.shape[1] when the utterance ran zero-padded in its length bucket: the frames past N are not its own
            pseudo_targets = decoder(log_p[-1, :N])
            ids = tokenizer(pseudo_targets).input_ids
            S = len(ids)
            targets = torch.tensor([ids if S else [0]] * num_negatives, dtype=torch.int32, device=device)
            aug = log_p[:num_negatives].contiguous()
            B = aug.shape[0]
            ilen = torch.full((B,), N, dtype=torch.int32, device=device)                         # CTC over the utterance's own frames; zero gradient past them
            tlen = torch.full((B,), S, dtype=torch.int32, device=device)
            _, _, g_lp = ops.ctc_loss(aug, targets, ilen, tlen, blank, reduction="mean", grad_scale=1.0)   # lib.py:351,434
            g_logits = ops.log_softmax_bwd(aug, g_lp)
            model.backward(g_logits, n_active=num_negatives)                                     # loss.backward(), lib.py:438
            ops.clip_grad_norm(model.flat_grads, 10.0)                                           # lib.py:442
            optimizer.step()
            optimizer.zero_grad()
            utterances[idx]['probs'] = log_p[-1, :N].detach().cpu()
    model.flat_params.copy_(original)                                                            # lib.py:459-460
    model.use_graphs = was_graphs
    return utterances


def dynamic_eval_ctc_loss(args, model, spec, seq_len, overlap, tokenizer, processor, use_tqdm=True, optim=MADGRAD, num_negatives=1,
                          lr_args={'lr': 1e-9}, return_device=False):
    """reference wav2vec2/lib.py:41-235: waveform windows (`-seq 131072 -o 0`), online stitching of exp(log_p[-1])."""
    device = model.device
    spec = spec.to(device=device, dtype=torch.float32)                                           # [1, L] waveform
    spec_n = spec.shape[-1]
    downsampling_factor = 4
    original = _snapshot(model)
    optimizer = optim(model.parameters(), **lr_args)
    blank = tokenizer.blank_id
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=blank, device=device)
    if seq_len > spec_n:
        seq_len, overlap = spec_n, 0
    assert overlap / downsampling_factor == overlap // downsampling_factor, 'Overlap must be a multiple of the downsampling factor'
    V = tokenizer.vocab_size
    acc = torch.zeros(spec_n // 4 + seq_len, V, device=device, dtype=torch.float32)
    cnt = torch.zeros(spec_n // 4 + seq_len, device=device, dtype=torch.float32)
    last_ulen, kill_next, training_data = None, False, {}
    for i in range(0, spec_n, seq_len - overlap):                                               # lib.py:116-126
        chunk = spec[:, i:i + seq_len]
        u_len = chunk.shape[-1]
        if kill_next:
            break
        elif last_ulen is not None and u_len < last_ulen:
            kill_next = True
        last_ulen = u_len
        training_data[i] = chunk
    outputs = {}
    was_graphs = model.use_graphs
    model.use_graphs = bool(args.__dict__.get('use_graphs', True))      # every full window falls into one length bucket: captured once, replayed
    model.bucket_frames = int(args.__dict__.get('bucket_frames', model.bucket_frames))
    for epoch in range(args.__dict__.get('epochs', 1)):
        outputs = {}
        keys = list(training_data.keys())
        keys = random.sample(keys, len(keys)) if args.__dict__.get('shuffle', False) else keys
        for i in (tqdm(keys) if use_tqdm else keys):
            chunk = training_data[i]
            u_len = chunk.shape[-1]
            audio = chunk.reshape(1, -1).repeat(num_negatives + 1, 1).contiguous()
            if args.__dict__.get('wav_augment', True):                                          # lib.py:144-156 on the first copies
                for j in range(num_negatives):
                    wav_augment_chunk(audio[j])
            input_values = normalize_waveform(audio)
            with torch.enable_grad():
                out = model(input_values)
            log_p = ops.log_softmax(out.logits)
            N = out.frames                                                                      # the window's own frames (see dynamic_eval_ctc_loss_su)
            ids = tokenizer(decoder(log_p[-1, :N])).input_ids
            S = len(ids)
            targets = torch.tensor([ids if S else [0]] * num_negatives, dtype=torch.int32, device=device)
            aug = log_p[:num_negatives].contiguous()
            B = aug.shape[0]
            ilen = torch.full((B,), N, dtype=torch.int32, device=device); tlen = torch.full((B,), S, dtype=torch.int32, device=device)
            _, _, g_lp = ops.ctc_loss(aug, targets, ilen, tlen, blank, reduction="sum", grad_scale=1.0 / (N * B))
            optimizer.zero_grad()
            model.backward(ops.log_softmax_bwd(aug, g_lp), n_active=num_negatives)
            optimizer.step()
            ds_len = N
            outputs[i] = (log_p[-1, :N], ds_len, int(overlap / (u_len / ds_len)))
    pos = end = 0
    for i in sorted(outputs):
        lp, ds_len, ov = outputs[i]
        pos -= ov if i != 0 else 0
        ops.stitch_accumulate(lp, acc, cnt, pos)
        pos += ds_len
        end = max(end, pos)
    logits = ops.stitch_finalize(acc, cnt, end)
    model.flat_params.copy_(original)
    model.use_graphs = was_graphs
    return logits if return_device else logits.cpu().numpy()


dynamic_eval = dynamic_eval_ctc_loss
dynamic_eval_su = dynamic_eval_ctc_loss_su


def apply_args(parser, argv=None):
    """reference wav2vec2/lib.py:477-493 (same flags).  The reference falls back to `paths.checkpoints.wav2vec2` when -c is empty;
    no paths.yaml / hub access exists offline, so an empty -c means seeded weights of the base-960h architecture."""
    parser.add_argument('-c', '--checkpoint', type=str, default='', help='path to checkpoint')
    parser.add_argument('-split', '--split', type=str, default='test', help='test or dev split')
    parser.add_argument('-seq', '--seq_len', type=int, default=131072)
    parser.add_argument('-overlap', '--overlap', type=int, default=0)
    parser.add_argument('-nv', '--not_verbose', action='store_true', help='verbose')
    parser.add_argument('-log', '--log', type=str, default='')
    parser.add_argument('-shuffle', '--shuffle', action='store_true', help='shuffle')
    parser.add_argument('-epochs', '--epochs', type=int, default=1, help='epochs')
    parser.add_argument('-dfa', '--disable_flash_attention', action='store_true', help='disable flash attention')
    args = parser.parse_args(argv)
    args.verbose = not args.not_verbose
    return args
