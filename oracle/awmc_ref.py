"""ORACLE (test infrastructure only).  CPU restatement of the reference's AWMC loop (reference lcasr/lib.py:206-376)
in plain PyTorch: anchor EMA (decay 1.0) and leader EMA (decay `ema_decay`) teachers, noisy student, CTC against both
label banks / (N*B*2), optimiser step, leader update, clean forward after the last epoch, stitch as lib.py:351-365.
`torch_ema.ExponentialMovingAverage` is an un-vendored dependency (absent here; unpinned in the reference): restated from
its published behaviour — shadow -= (1 - d)(shadow - p), d = min(decay, (1 + n)/(10 + n)) with n counted from 1, and
`average_parameters()` swapping the shadow in and the live weights back out.  PARITY UNPINNED against torch_ema."""
import contextlib

import numpy as np
import torch

from .dynamic_eval_ref import apply_masks, draw_masks, greedy_ctc_ids, prepare_chunks


class EMARef:
    def __init__(self, params, decay):
        self.params = list(params)
        self.decay, self.num_updates = decay, 0
        self.shadow = [p.clone().detach() for p in self.params]

    def update(self):
        self.num_updates += 1
        d = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            for s, p in zip(self.shadow, self.params):
                s.sub_((1.0 - d) * (s - p))

    @contextlib.contextmanager
    def average_parameters(self):
        saved = [p.clone().detach() for p in self.params]
        with torch.no_grad():
            for p, s in zip(self.params, self.shadow):
                p.copy_(s)
        try:
            yield
        finally:
            with torch.no_grad():
                for p, s in zip(self.params, saved):
                    p.copy_(s)


def bitfit_ref(model):
    """reference lcasr/lib.py:148-160: requires_grad False everywhere, then True for the biases of LayerNorm / Linear /
    BatchRenorm modules.  On the oracle's module tree (oracle/conformer_ref.py): `_Norm` of kind layer_norm / batch_renorm, and
    the `_Lin` modules that are nn.Linear upstream (attn.qkv, attn.out, subsampling.out, decoder.ff, decoder.reproj; the
    pointwise convolutions of the conv module and of the subsampling are Conv modules upstream)."""
    for p in model.parameters():
        p.requires_grad = False
    linear = ("attn.qkv", "attn.out", "subsampling.out", "decoder.ff", "decoder.reproj")
    for name, module in model.named_modules():
        cls = type(module).__name__
        if cls == "_Norm" and module.bias is not None:
            module.bias.requires_grad = True
        if cls == "_Lin" and module.bias is not None and name.endswith(linear):
            module.bias.requires_grad = True
    return model


def awmc_ref(model, spec, seq_len, overlap, tokenizer, optimizer_cls, lr_args, spec_augment_config, epochs=1, ema_decay=0.999,
             downsampling_factor=8, fixed_masks=None, return_params=False, bitfit=False):
    spec_n = spec.shape[-1]
    original = [p.clone().detach().cpu() for p in model.parameters()]
    if bitfit:                                            # reference lib.py:234-235
        bitfit_ref(model)
    model.train()
    ema_leader = EMARef(model.parameters(), ema_decay); ema_leader.update()
    ema_anchor = EMARef(model.parameters(), 1.0); ema_anchor.update()
    blank = model.decoder.num_classes - 1
    ctc_loss_fn = torch.nn.CTCLoss(blank=blank, reduction='sum')
    optimizer = optimizer_cls(model.parameters(), **lr_args)
    if seq_len > spec_n:
        seq_len, overlap = spec_n, 0
    assert overlap / downsampling_factor == overlap // downsampling_factor
    C = tokenizer.vocab_size() + 1
    all_logits, logit_count = torch.zeros((1, spec_n // 4 + seq_len, C)), torch.zeros((1, spec_n // 4 + seq_len, C))
    training_data, training_keys = prepare_chunks(spec, seq_len, overlap)
    model_outputs = {}
    model.eval()

    def targets_of(chunk):
        out = model(audio_signal=chunk)
        ids = greedy_ctc_ids(out['final_posteriors'][-1].detach(), blank)
        return torch.LongTensor(tokenizer.encode(tokenizer.decode(ids))).unsqueeze(0).transpose(0, 1)

    for i in training_keys:
        label_bank = [None, None]
        for j in range(epochs):
            audio_chunk = training_data[i].clone()
            if j == 0:
                with ema_anchor.average_parameters(), torch.no_grad():
                    label_bank[0] = targets_of(audio_chunk)
            with ema_leader.average_parameters(), torch.no_grad():
                label_bank[1] = targets_of(audio_chunk)
            F_, u_len = audio_chunk.shape[1], audio_chunk.shape[-1]
            if fixed_masks is not None:
                masks = fixed_masks[i]
            else:
                masks = (draw_masks(spec_augment_config.get('n_freq_masks', 0), spec_augment_config.get('freq_mask_param', 42), F_), ([], []))
            apply_masks(audio_chunk[0], masks, spec_augment_config.get('zero_masking', False))
            out = model(audio_signal=audio_chunk)
            labels = [el for el in label_bank if el.shape[0] > 0]
            lens = torch.LongTensor([el.shape[0] for el in labels])
            if len(labels) == 0:
                labels = [torch.LongTensor([[]]).T]
                lens = torch.LongTensor([0])
            labels = torch.nn.utils.rnn.pad_sequence(sequences=labels, batch_first=False, padding_value=0)
            labels = labels.squeeze(2).transpose(0, 1)
            N, B = out['final_posteriors'].shape[1], out['final_posteriors'].shape[0]
            loss = ctc_loss_fn(out['final_posteriors'].repeat(lens.shape[0], 1, 1).transpose(0, 1), targets=labels,
                               input_lengths=torch.LongTensor([N] * labels.shape[0]), target_lengths=lens) / (N * B * 2)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            ema_leader.update()
            if j == epochs - 1:
                audio_chunk = training_data[i].clone()
                with torch.no_grad():
                    out = model(audio_signal=audio_chunk)
                logits = torch.exp(out['final_posteriors'][0].detach())
                ds_len = logits.shape[-2]
                model_outputs[i] = {'logits': logits, 'ds_len': ds_len, 'overlap_ds': int(overlap / (audio_chunk.shape[-1] / ds_len))}
    pos = 0
    for i in sorted(model_outputs):
        lg, ds_len, ov = model_outputs[i]['logits'], model_outputs[i]['ds_len'], model_outputs[i]['overlap_ds']
        pos -= ov if i != 0 else 0
        logit_count[:, pos:pos + ds_len, :] += 1
        all_logits[:, pos:pos + ds_len, :] += lg
        pos += ds_len
    B, N, C = all_logits.shape
    keep = logit_count.sum(dim=-1) != 0
    logits = torch.log(all_logits[keep].reshape(B, -1, C) / logit_count[keep].reshape(B, -1, C))
    if return_params:
        updated = [p.clone().detach().cpu() for p in model.parameters()]
    for p, po in zip(model.parameters(), original):
        p.data = po.data.to(p.device)
        p.requires_grad = True
    out = logits.squeeze(0).numpy().astype(np.float32)
    return (out, updated) if return_params else out
