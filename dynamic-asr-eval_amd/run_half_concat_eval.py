"""`concatenate_specs` and `adapt_on_concat_only` with the reference's signatures
(reference lcasr/run_half_concat_eval.py:58-61,64-160): Loop A of dynamic eval over a concatenated spectrogram without
building the stitched logits; returns the adapted parameters as CPU clones and restores the model.  Used by the
whole-concat harness (reference lcasr/run_whole_concat_eval.py:123-152, BASELINE config 4)."""
import random

import torch

from . import lib, ops
from .augment import SpecAugment
from .awmc import AWMC
from .decoding import GreedyCTCDecoder
from .optim import MADGRAD

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, **_):
        return x


def concatenate_specs(specs):
    if len(specs) == 0:
        raise ValueError('Cannot concatenate an empty list of spectrograms.')
    return torch.cat(specs, dim=-1)


def adapt_on_concat_only(args, model, concat_spec, tokenizer, beamsearch=None, adapt_overlap=None):
    if getattr(args, 'awmc', False):
        _, updated = AWMC(args, model, concat_spec, args.seq_len, adapt_overlap, tokenizer, use_tqdm=False,
                          beam_search_fn=beamsearch, return_params=True, return_device=True)
        return updated
    if beamsearch is not None and args.__dict__.get('lm_tta_beams', 3) != 0:
        lib._unsupported("LM beam-search pseudo-labels")
    if not lib._is_native(model):
        raise ops.DynError("adapt_on_concat_only: only the HIP SCConformerXL is supported")
    device = model.device
    spec_n = concat_spec.shape[-1]
    downsampling_factor = args.config['model']['subsampling_factor']
    seq_len = args.seq_len if args.seq_len != -1 else args.config['audio_chunking']['size']
    spec_augment_config = lib.get_specaugment_config_from_args(args)
    lr_args = lib.get_lr_args_from_args(args)
    fs = lib.get_frame_shuffle_config_from_args(args)
    if args.__dict__.get('random_noise', 0.0) or fs['time_dimension'] or fs['freq_dimension'] or \
            lib.get_cutout_params_from_args(args, seq_len)['num_rectangles'] or args.__dict__.get('entropy_augmentation_enabled', False):
        lib._unsupported("random_noise / frame_shuffle / cutout / entropy_augmentation")
    num_negatives = 1
    original_flat = model.flat_params.clone()
    num_classes = model.decoder.num_classes
    blank = num_classes - 1
    optimizer = MADGRAD(model.parameters(), **lr_args)            # hard-wired in the reference (:98)
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=blank, device=device)
    augmentation = SpecAugment(**spec_augment_config)
    fixed_masks = args.__dict__.get('spec_augment_fixed_masks', None)
    if seq_len > spec_n:
        seq_len, adapt_overlap = spec_n, 0
    else:
        adapt_overlap = adapt_overlap if adapt_overlap != -1 else args.config['audio_chunking']['overlap']
    assert args.config['training'].get('max_seq_len', 0) == 0, 'caching is not used anymore'
    assert adapt_overlap / downsampling_factor == adapt_overlap // downsampling_factor, 'Overlap must be a multiple of the downsampling factor'
    epochs = args.__dict__.get('epochs', 1)
    shuffle = args.__dict__.get('shuffle', False)
    model.eval()
    spec_dev = concat_spec.to(device=device, dtype=torch.float32)
    Fq = spec_dev.shape[1]
    training_data, training_keys = lib.prepare_chunks(spec_dev, seq_len, adapt_overlap)
    if not args.__dict__.get('quiet', False):
        print(f'Adapt-only pass on concatenated spec: {len(training_keys)} chunks, seq_len={seq_len}, overlap={adapt_overlap}')
    for epoch in range(epochs):
        cur_keys = list(training_keys)
        cur_keys = random.sample(cur_keys, len(cur_keys)) if shuffle else cur_keys
        for i in cur_keys:
            view = training_data[i][0]
            u_len = view.shape[-1]
            audio_chunk = torch.empty(num_negatives + 1, Fq, u_len, device=device, dtype=torch.float32)
            for b in range(num_negatives + 1):
                audio_chunk[b].copy_(view)
            for b in range(num_negatives):
                masks = fixed_masks[i] if fixed_masks is not None else augmentation.draw(Fq, u_len)
                if masks[0][0] or masks[1][0]:
                    augmentation.apply(audio_chunk[b], masks, lib._window_fill_value(audio_chunk[b], augmentation.zero_masking))
            with torch.enable_grad():
                post = model(audio_signal=audio_chunk)['final_posteriors']
            target_ids = tokenizer.encode(decoder(post[-1].detach()))
            S = len(target_ids)
            targets = torch.tensor([target_ids if S else [0]] * num_negatives, dtype=torch.int32, device=device)
            aug = post[:num_negatives]
            n_tokens, batch_size = aug.shape[1], aug.shape[0]
            ilen = torch.full((batch_size,), n_tokens, dtype=torch.int32, device=device)
            tlen = torch.full((batch_size,), S, dtype=torch.int32, device=device)
            _, _, g = ops.ctc_loss(aug.contiguous(), targets, ilen, tlen, blank, reduction="sum",
                                   grad_scale=1.0 / (n_tokens * batch_size))
            optimizer.zero_grad()
            model.backward(g, n_active=num_negatives)
            optimizer.step()
    updated_model_params = [p.clone().detach().cpu() for p in model.parameters()]
    model.flat_params.copy_(original_flat)
    return updated_model_params
