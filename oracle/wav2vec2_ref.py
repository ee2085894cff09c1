"""ORACLE (test infrastructure only).  CPU restatements of the reference's wav2vec2 dynamic-eval loops: the chunked
`dynamic_eval_ctc_loss` (reference wav2vec2/lib.py:41-235, see dynamic_eval_chunked_ref below) and the per-utterance loop
`dynamic_eval_ctc_loss_su` (reference wav2vec2/lib.py:293-462) on the transformers CPU model (the class the reference
itself loads, lib.py:20-23): snapshot weights (:314-315), CTCLoss(blank, reduction='mean') (:351), fresh optimiser (:354),
per utterance: batch of num_negatives+1 identical waveforms (:385; the augmentation lines are commented out in the
reference), HF feature-extractor normalisation = zero mean / unit variance with eps 1e-7 (:406), forward (:413),
log_softmax (:417), greedy pseudo-label of the last copy (:419), tokenise (:430), CTC loss on the first copies (:434),
backward (:438), clip_grad_norm_(10) + step + zero_grad (:441-444), keep log_p[-1] (:455-456); restore (:459-460).
The HF processor object itself needs downloaded files; its normalisation rule is restated (transformers
Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm)."""
import random

import torch
import torch.nn.functional as F

from .dynamic_eval_ref import greedy_ctc_ids


def normalize_waveform(x):
    """[B, L]: (x - mean) / sqrt(var + 1e-7) per row, biased variance (Wav2Vec2FeatureExtractor)."""
    return (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-7)


def dynamic_eval_su_ref(args, model, utterances, tokenizer, optimizer_cls, num_negatives=1, lr_args=None):
    original = [p.clone().detach() for p in model.parameters()]
    ctc_loss_fn = torch.nn.CTCLoss(blank=tokenizer.blank_id, reduction='mean')
    optimizer = optimizer_cls(model.parameters(), **(lr_args or {'lr': 1e-15}))
    for epoch in range(args.__dict__.get('epochs', 1)):
        indexes = list(range(len(utterances)))
        indexes = random.sample(indexes, len(indexes)) if args.__dict__.get('shuffle', False) else indexes
        for idx in indexes:
            audio = utterances[idx]['waveform'].repeat(num_negatives + 1, 1, 1).contiguous().squeeze(1)
            input_values = normalize_waveform(audio)
            logits = model(input_values).logits
            log_p = F.log_softmax(logits, dim=-1)
            pseudo = tokenizer.decode(greedy_ctc_ids(log_p[-1].detach(), tokenizer.blank_id))
            targets = torch.LongTensor(tokenizer(pseudo).input_ids).unsqueeze(0).repeat(num_negatives, 1)
            aug = log_p[:num_negatives]
            N, B = aug.shape[1], aug.shape[0]
            loss = ctc_loss_fn(aug.transpose(0, 1), targets, torch.LongTensor([N] * B), torch.LongTensor([targets.shape[1]] * B))
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
            optimizer.step()
            optimizer.zero_grad()
            utterances[idx]['probs'] = log_p[-1].detach().cpu()
    for p, po in zip(model.parameters(), original):
        p.data = po.data
    return utterances


def wav_augment_ref(x, n_rounds=100, max_seconds=0.1, rate=16000):
    """x [1, L]: 100 x WavAugment time_dropout(0.1 s) (zero a span of np.random.randint(0, 1600) samples starting at
    np.random.randint(0, max(1, L - length))), then additive_noise(zero noise, snr=0) = 0.5 * x (reference wav2vec2/lib.py:144-156).  `augment`
    is un-vendored (restated from the published effects, PARITY UNPINNED); the chain's last effect, `.reverb(50, 50, 100)` (sox), is not
    reproduced on either side."""
    import numpy as np
    max_frames = int(rate * max_seconds)
    for _ in range(n_rounds):
        length = int(np.random.randint(0, max_frames))
        start = int(np.random.randint(0, max(1, x.shape[-1] - length)))
        x[:, start:start + length] = 0.0
    return x * 0.5


def dynamic_eval_chunked_ref(args, model, spec, seq_len, overlap, tokenizer, optimizer_cls, num_negatives=1, lr_args=None, wav_augment=True):
    """reference wav2vec2/lib.py:41-235 on the transformers CPU model, statement by statement:
    downsampling_factor hard-coded 4 (:59), snapshot (:63-64), CTCLoss(blank, reduction='sum') (:66), fresh optimiser (:98),
    `seq_len > spec_n` rule (:104-105), overlap % 4 assert (:107), accumulators of spec_n // 4 + seq_len rows (:110), the inlined
    window rule (:116-126), per epoch a fresh model_outputs (:133); per window: num_negatives + 2 copies (:141), feature-extractor
    normalisation (:161), forward of all copies but the last (:163), log_softmax (:169), greedy pseudo-label of log_p[-1] (:170),
    tokenise (:174), CTC of the first num_negatives copies / (N * B) (:177-181), zero_grad / backward / step (:193-197), keep
    exp(log_p[-1]) with ds_len and overlap_ds = int(overlap / (u_len / ds_len)) (:205-210); stitch (:214-230); restore (:233-234).
    The WavAugment chain on the first copies (:144-156): 100 x time_dropout(0.1 s) and additive_noise(zeros, snr=0) are restated
    (`wav_augment_ref`; `augment` = facebookresearch/WavAugment is un-vendored: PARITY UNPINNED), `.reverb(50, 50, 100)` (sox) is not."""
    from .dynamic_eval_ref import stitch_ref
    spec_n = spec.shape[-1]
    downsampling_factor = 4
    original = [p.clone().detach() for p in model.parameters()]
    ctc_loss_fn = torch.nn.CTCLoss(blank=tokenizer.blank_id, reduction='sum')
    optimizer = optimizer_cls(model.parameters(), **(lr_args or {'lr': 1e-9}))
    if seq_len > spec_n:
        seq_len, overlap = spec_n, 0
    assert overlap / downsampling_factor == overlap // downsampling_factor, 'Overlap must be a multiple of the downsampling factor'
    V = tokenizer.vocab_size
    all_logits, logit_count = torch.zeros((1, spec_n // 4 + seq_len, V)), torch.zeros((1, spec_n // 4 + seq_len, V))
    last_ulen, kill_next = None, False
    training_data = {}
    for i in range(0, spec_n, seq_len - overlap):
        audio_chunk = spec[:, i:i + seq_len]
        u_len = audio_chunk.shape[-1]
        if kill_next:
            break
        elif last_ulen is not None and u_len < last_ulen:
            kill_next = True
        last_ulen = u_len
        training_data[i] = audio_chunk
    model_outputs = {}
    for epoch in range(args.__dict__.get('epochs', 1)):
        model_outputs = {}
        training_keys = list(training_data.keys())
        training_keys = random.sample(training_keys, len(training_keys)) if args.__dict__.get('shuffle', False) else training_keys
        for i in training_keys:
            audio_chunk = training_data[i].clone()
            audio_chunk = audio_chunk.repeat(num_negatives + 2, 1, 1).contiguous()
            if wav_augment:
                for j in range(num_negatives):
                    audio_chunk[j] = wav_augment_ref(audio_chunk[j])
            u_len = audio_chunk.shape[-1]
            audio_chunk = audio_chunk.squeeze(1)
            input_values = normalize_waveform(audio_chunk)
            logits = model(input_values[:-1]).logits
            log_p = F.log_softmax(logits, dim=-1)
            pseudo = tokenizer.decode(greedy_ctc_ids(log_p[-1].detach(), tokenizer.blank_id))
            targets = torch.LongTensor(tokenizer(pseudo).input_ids).unsqueeze(0).repeat(num_negatives, 1)
            aug = log_p[:num_negatives]
            N, B = aug.shape[1], aug.shape[0]
            loss = ctc_loss_fn(aug.transpose(0, 1), targets, torch.LongTensor([N] * B), torch.LongTensor([targets.shape[1]] * B)) / (N * B)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            probs = torch.exp(log_p[-1].detach())
            ds_len = probs.shape[-2]
            model_outputs[i] = {'logits': probs, 'ds_len': ds_len, 'overlap_ds': int(overlap / (u_len / ds_len))}
    logits = stitch_ref(model_outputs, all_logits, logit_count)
    for p, po in zip(model.parameters(), original):
        p.data = po.data
    return logits.squeeze(0).numpy()
