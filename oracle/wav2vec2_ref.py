"""ORACLE (test infrastructure only).  CPU restatement of the reference's per-utterance wav2vec2 dynamic-eval loop
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
