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


def _su_gen(args, model, utterances, seq_len, overlap, tokenizer, processor, optim=MADGRAD, num_negatives=1, lr_args={'lr': 1e-15},
            ngram_decoder=None, use_tqdm=False):
    """Generator form of dynamic_eval_ctc_loss_su (reference wav2vec2/lib.py:293-462): yields where the host would otherwise block on the GPU
    (the pseudo-label ids of every utterance, the log-probs at the end), so that dynamic_eval_su_many can keep several recordings in flight
    from one host thread.  Nothing in the loop blocks the host except those two waits: the waveforms go up in ONE pinned transfer, the label
    ids come down and go up through pinned buffers, the utterances' log-probs collect in one pinned arena."""
    if ngram_decoder is not None:
        raise NotImplementedError("n-gram (pyctcdecode) pseudo-labels need the un-vendored decoder and its ARPA file")
    device = model.device
    downsampling_factor = 4
    original = _snapshot(model)
    blank = tokenizer.blank_id
    optimizer = optim(model.parameters(), **lr_args)
    assert overlap / downsampling_factor == overlap // downsampling_factor, 'Overlap must be a multiple of the downsampling factor'
    model = disable_dropout(model)
    # hipGraph replay over utterance-length buckets (wav2vec2_model.py::forward): `args.use_graphs` (default on), `args.bucket_frames`
    was_graphs = model.use_graphs
    model.use_graphs = bool(args.__dict__.get('use_graphs', True))
    model.bucket_frames = int(args.__dict__.get('bucket_frames', model.bucket_frames))
    lens = [int(u['waveform'].shape[-1]) for u in utterances]
    offs, total = [], 0
    for L in lens:
        offs.append(total)
        total += (L + 3) // 4 * 4                                                                # 16-byte aligned rows
    host = torch.empty(max(total, 4), dtype=torch.float32, pin_memory=True)
    for u, o, L in zip(utterances, offs, lens):
        host[o:o + L] = u['waveform'].reshape(-1).to(torch.float32)
    waves = host.to(device, non_blocking=True)
    frames = [model.conv_lengths(L)[-1] for L in lens]
    foffs = [sum(frames[:i]) for i in range(len(frames))]
    probs_host, pinned, tgt_ring, tgt_turn = None, None, None, 0
    for epoch in range(args.__dict__.get('epochs', 1)):
        indexes = list(range(len(utterances)))
        indexes = random.sample(indexes, len(indexes)) if args.__dict__.get('shuffle', False) else indexes
        pbar = tqdm(indexes) if use_tqdm else indexes
        for idx in pbar:
            L = lens[idx]
            audio_chunk = waves[offs[idx]:offs[idx] + L].view(1, L).repeat(num_negatives + 1, 1)     # [B, L]; the copies stay clean (lib.py:391-412)
            input_values = normalize_waveform(audio_chunk)
            with torch.enable_grad():
                out = model(input_values)
            log_p = ops.log_softmax(out.logits)                                                  # F.log_softmax, lib.py:417
            N = out.frames                      # < log_p.shape[1] when the utterance ran zero-padded in its length bucket: the frames past N are not its own
            ids_dev, n_dev = ops.ctc_greedy(log_p[-1, :N], blank)                                # decoder(log_p[-1]), lib.py:419: only the ids cross PCIe
            if pinned is None or pinned[0].shape[1] < ids_dev.shape[1]:
                pinned = (torch.empty(1, 2 * ids_dev.shape[1], dtype=torch.int32, pin_memory=True), torch.empty(1, dtype=torch.int32, pin_memory=True))
            pinned[0][:, :ids_dev.shape[1]].copy_(ids_dev, non_blocking=True)
            pinned[1].copy_(n_dev, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
            yield                                                                                # another recording may use the host meanwhile
            ready.synchronize()
            pseudo_targets = tokenizer.decode(pinned[0][0, :int(pinned[1][0])].tolist())
            ids = tokenizer(pseudo_targets).input_ids
            S = len(ids)
            if tgt_ring is None or tgt_ring[0].shape[1] < max(S, 1):
                tgt_ring = [torch.empty(num_negatives, max(2 * S, 256), dtype=torch.int32, pin_memory=True) for _ in range(4)]
            slot = tgt_ring[tgt_turn % 4]                                                        # reused 4 utterances later: its upload is long done
            tgt_turn += 1
            slot[:, :max(S, 1)] = torch.as_tensor(ids if S else [0], dtype=torch.int32)
            targets = torch.empty(num_negatives, max(S, 1), dtype=torch.int32, device=device)
            targets.copy_(slot[:, :max(S, 1)], non_blocking=True)
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
            if probs_host is None:
                probs_host = torch.empty(max(sum(frames), 1), log_p.shape[-1], dtype=torch.float32, pin_memory=True)
            probs_host[foffs[idx]:foffs[idx] + N].copy_(log_p[-1, :N], non_blocking=True)        # utterances[idx]['probs'], lib.py:452
    model.flat_params.copy_(original)                                                            # lib.py:459-460
    model.use_graphs = was_graphs
    done = torch.cuda.Event()
    done.record()
    yield
    done.synchronize()
    if probs_host is not None:
        for idx in range(len(utterances)):
            utterances[idx]['probs'] = probs_host[foffs[idx]:foffs[idx] + frames[idx]].clone()
    return utterances


def dynamic_eval_ctc_loss_su(args, model, utterances, seq_len, overlap, tokenizer, processor, use_tqdm=True, optim=MADGRAD,
                             num_negatives=1, lr_args={'lr': 1e-15}, ngram_decoder=None):
    """reference wav2vec2/lib.py:293-462 (one recording's utterances, weights carried from utterance to utterance and restored at the end)."""
    gen = _su_gen(args, model, utterances, seq_len, overlap, tokenizer, processor, optim, num_negatives, lr_args, ngram_decoder, use_tqdm)
    try:
        while True:
            next(gen)
    except StopIteration as stop:
        return stop.value


def dynamic_eval_su_many(args, models, utterance_lists, seq_len, overlap, tokenizer, processor, optim=MADGRAD, num_negatives=1,
                         lr_args={'lr': 1e-15}):
    """Several RECORDINGS in flight on one GPU from one host thread (the `chains` of the conformer path, lib.dynamic_eval_many): the reference's
    driver calls dynamic_eval_su once per talk and every call starts from the checkpoint's weights (wav2vec2/tedlium/run.py:155,
    lib.py:455-460), so talks are independent.  One utterance step of wav2vec2-base is ~200 matrix products of 40 - 500 tiles each
    (profiles/r04_kernel_stats_wav2vec2_su.csv): alone they leave most of the 256 CUs idle; the steps of other talks fill them.  Each model
    replica in `models` owns a HIP stream and runs one talk at a time; returns the utterance lists in order."""
    from .lib import _CHAIN_STREAMS, _new_chain_stream
    device = models[0].device
    key = torch.device(device).index
    while len(_CHAIN_STREAMS.setdefault(key, [])) < len(models):
        _CHAIN_STREAMS[key].append(_new_chain_stream(device, len(_CHAIN_STREAMS[key])))
    streams = _CHAIN_STREAMS[key][:len(models)]
    main = torch.cuda.current_stream(device)
    for st in streams:
        st.wait_stream(main)
    pending = list(enumerate(utterance_lists))
    results = [None] * len(utterance_lists)
    free, active = list(range(len(models)))[::-1], []
    while pending or active:
        while pending and free:
            ci = free.pop()
            idx, utts = pending.pop(0)
            active.append([_su_gen(args, models[ci], utts, seq_len, overlap, tokenizer, processor, optim, num_negatives, lr_args), ci, idx])
        for item in list(active):
            gen, ci, idx = item
            with torch.cuda.stream(streams[ci]):
                try:
                    next(gen)
                except StopIteration as stop:
                    results[idx] = stop.value
                    active.remove(item)
                    free.append(ci)
    for st in streams:
        main.wait_stream(st)
    return results


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
