"""ORACLE (test infrastructure only — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product path).

CPU restatement, in plain PyTorch fp32, of the reference's dynamic-eval inner loop
`dynamic_eval_ctc_loss` (reference lcasr/lib.py:450-640), following it line by line:
  prepare_chunks                      lib.py:128-145
  snapshot of the weights             lib.py:482-483
  CTCLoss(blank=V, reduction='sum')   lib.py:492
  fresh optimiser per call            lib.py:494
  per window: batch of 2 (copy 0 augmented, copy -1 clean)        lib.py:538-545
              forward, greedy pseudo-label of the clean copy      lib.py:550,559
              tokenizer.encode(text) -> targets                   lib.py:569
              CTC loss of the augmented copy / (N*B)              lib.py:572-575
              zero_grad / backward / step                         lib.py:578-581
              online: keep exp(clean log-probs)                   lib.py:583-589
  offline final pass (no_grad, B=1)   lib.py:594-612
  stitch: exp, overlap-add, count, divide, log                    lib.py:615-629
  restore the weights                 lib.py:636-637
The reference itself cannot be imported here: `import lib` fails with ModuleNotFoundError (omegaconf, lcasr, lming,
torch_ema, whisper, madgrad are un-vendored and absent).  The pieces it takes from those packages are
restated in this directory: the acoustic model (conformer_ref.py), MADGRAD (madgrad_ref.py), greedy CTC decoding and
the SpecAugment mask rule (below).  torch.nn.CTCLoss and autograd are the very ops the reference calls.
PARITY UNPINNED against the upstream model / SpecAugment / MADGRAD (the reference holds no fixture for them);
pinned for windowing, stitching arithmetic and the loop structure by the reference source cited above.
"""
import random

import numpy as np
import torch


def prepare_chunks(spec, seq_len, overlap):
    """reference lcasr/lib.py:128-145"""
    spec_n = spec.shape[-1]
    last_ulen, kill_next = None, False
    if spec_n <= seq_len:
        return {0: spec}, [0]
    training_data = {}
    for i in range(0, spec_n, seq_len - overlap):
        audio_chunk = spec[:, :, i:i + seq_len]
        u_len = audio_chunk.shape[-1]
        if kill_next:
            break
        elif last_ulen is not None and u_len < last_ulen:
            kill_next = True
        last_ulen = u_len
        training_data[i] = audio_chunk
    return training_data, list(training_data.keys())


def greedy_ctc_ids(log_probs, blank_id):
    """argmax -> collapse repeats -> drop blank (upstream GreedyCTCDecoder, call sites lib.py:498,559,565)."""
    ids = torch.argmax(log_probs, dim=-1).tolist()
    out, prev = [], None
    for i in ids:
        if i != blank_id and i != prev:
            out.append(i)
        prev = i
    return out


def draw_masks(n_masks, param, size, generator=None):
    """Same rule as the product's augment.draw_masks (defined by this build, see its docstring)."""
    starts, widths = [], []
    for _ in range(int(n_masks)):
        w = int(torch.randint(0, int(param) + 1, (1,), generator=generator).item()) if param > 0 else 0
        w = min(w, size)
        s = int(torch.randint(0, size - w + 1, (1,), generator=generator).item())
        starts.append(s)
        widths.append(w)
    return starts, widths


def apply_masks(window, masks, zero_masking):
    """window [F, T] (modified in place)."""
    (f0, fw), (t0, tw) = masks
    if not (f0 or t0):
        return window
    fill = 0.0 if zero_masking else float(window.mean())
    for s, w in zip(f0, fw):
        window[s:s + w, :] = fill
    for s, w in zip(t0, tw):
        window[:, s:s + w] = fill
    return window


def stitch_ref(model_outputs, all_logits, logit_count):
    """The stitch of reference lcasr/lib.py:615-629 (the same statements close run_seq_eval.py:130-144 and
    run_within_recording_loo_eval.py): windows in key order, `logit_position -= overlap_ds` except for key 0, overlap-add
    of the probabilities and of a coverage count, rows never covered dropped, log(sum / count).
    model_outputs: {key: {'logits': probs [1, ds_len, C] or [ds_len, C], 'ds_len', 'overlap_ds'}}; the two accumulators are
    zero tensors [1, rows, C].  Pinned by tests/golden/reference_pins.npz (the reference's statements executed unchanged)."""
    logit_position = 0
    for i in sorted(list(model_outputs.keys())):
        logits, ds_len, overlap_ds = model_outputs[i]['logits'], model_outputs[i]['ds_len'], model_outputs[i]['overlap_ds']
        logit_position -= overlap_ds if i != 0 else 0
        logit_count[:, logit_position:logit_position + ds_len, :] += 1
        all_logits[:, logit_position:logit_position + ds_len, :] += logits
        logit_position += ds_len
    B, N, C = all_logits.shape
    all_logits = all_logits[logit_count.sum(dim=-1) != 0].reshape(B, -1, C)
    logit_count = logit_count[logit_count.sum(dim=-1) != 0].reshape(B, -1, C)
    return torch.log(all_logits / logit_count)


def dynamic_eval_ref(model, spec, seq_len, overlap, tokenizer, optimizer_cls, lr_args, spec_augment_config, epochs=1,
                     shuffle=False, online=False, downsampling_factor=8, fixed_masks=None, return_params=False,
                     max_windows=None, timings=None, also_online=False):
    """`max_windows` (benchmarks only) stops the adaptation loop after that many windows.
    `timings` (benchmarks only): a dict that receives 'adapt' / 'final' = seconds per window of the two loops.
    `also_online` (tests only, offline mode): the adaptation loop is the same in both modes, so one run can also hand back what
    `online=True` would have stitched (the clean copy's posteriors of every adapt step, lib.py:583-589) -> (offline, online[, params]).
    A float64 model / spectrogram runs the whole loop in float64 (the noise-floor reference of tests/drift_check.py)."""
    import time
    spec_n = spec.shape[-1]
    original_model_params = [p.clone().detach().cpu() for p in model.parameters()]
    num_negatives = 1
    blank = model.decoder.num_classes - 1
    ctc_loss_fn = torch.nn.CTCLoss(blank=blank, reduction='sum')
    optimizer = optimizer_cls(model.parameters(), **lr_args)
    if seq_len > spec_n:
        seq_len, overlap = spec_n, 0
    assert overlap / downsampling_factor == overlap // downsampling_factor
    all_logits = torch.zeros((1, spec_n // 4 + seq_len, tokenizer.vocab_size() + 1), dtype=spec.dtype)
    logit_count = torch.zeros((1, spec_n // 4 + seq_len, tokenizer.vocab_size() + 1), dtype=spec.dtype)
    loop_epochs = epochs                   # the reference's loop runs `range(args.epochs)` (lib.py:527) ...
    epochs = 1 if online else epochs       # ... although the printed count is forced to 1 in online mode (lib.py:515)
    shuffle = False if online else shuffle
    model_outputs = {}                     # online: overwritten by every epoch, so the LAST epoch is what gets stitched (lib.py:583-589)
    online_outputs = {}                    # also_online: what online=True would have kept
    model.eval()
    training_data, training_keys = prepare_chunks(spec, seq_len, overlap)
    n_done = 0
    for epoch in range(loop_epochs):
        training_keys = list(training_data.keys())
        training_keys = random.sample(training_keys, len(training_keys)) if shuffle else training_keys
        for i in training_keys:
            if max_windows is not None and n_done >= max_windows:
                break
            n_done += 1
            t_win = time.perf_counter()
            audio_chunk = training_data[i].clone()
            audio_chunk = audio_chunk.repeat(num_negatives + 1, 1, 1)
            F_, u_len = audio_chunk.shape[1], audio_chunk.shape[-1]
            for b in range(num_negatives):
                if fixed_masks is not None:
                    masks = fixed_masks[i]
                else:
                    fm = draw_masks(spec_augment_config.get('n_freq_masks', 0), spec_augment_config.get('freq_mask_param', 42), F_)
                    tp = spec_augment_config.get('time_mask_param', -1)
                    tp = tp if tp >= 0 else max(1, int(spec_augment_config.get('min_p', 0.05) * u_len))
                    masks = (fm, draw_masks(spec_augment_config.get('n_time_masks', 0), tp, u_len))
                apply_masks(audio_chunk[b], masks, spec_augment_config.get('zero_masking', False))
            out = model(audio_signal=audio_chunk)
            pseudo_ids = greedy_ctc_ids(out['final_posteriors'][-1].detach(), blank)
            pseudo_targets = tokenizer.decode(pseudo_ids)
            pseudo_targets = torch.LongTensor(tokenizer.encode(pseudo_targets)).unsqueeze(0).repeat(num_negatives, 1)
            augmented_outs = out['final_posteriors'][:num_negatives]
            N, B = augmented_outs.shape[1], augmented_outs.shape[0]
            loss = ctc_loss_fn(augmented_outs.transpose(0, 1), pseudo_targets, torch.LongTensor([N] * B),
                               torch.LongTensor([pseudo_targets.shape[1]] * B)) / (N * B)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            if online or also_online:
                logits = torch.exp(out['final_posteriors'][-1].detach())
                ds_len = logits.shape[-2]
                (model_outputs if online else online_outputs)[i] = {'logits': logits, 'ds_len': ds_len, 'overlap_ds': int(overlap / (u_len / ds_len))}
            if timings is not None:
                timings.setdefault('adapt', []).append(time.perf_counter() - t_win)
    if not online:
        model.eval()
        training_data, training_keys = prepare_chunks(spec, seq_len, overlap)
        for i in training_keys:
            t_win = time.perf_counter()
            audio_chunk = training_data[i].clone()
            u_len = audio_chunk.shape[-1]
            with torch.no_grad():
                out = model(audio_signal=audio_chunk)
            logits = torch.exp(out['final_posteriors'][0].detach())
            ds_len = logits.shape[-2]
            model_outputs[i] = {'logits': logits, 'ds_len': ds_len, 'overlap_ds': int(overlap / (u_len / ds_len))}
            if timings is not None:
                timings.setdefault('final', []).append(time.perf_counter() - t_win)
    logits = stitch_ref(model_outputs, all_logits, logit_count)
    if return_params:
        updated = [p.clone().detach().cpu() for p in model.parameters()]
    for p, p_orig in zip(model.parameters(), original_model_params):
        p.data = p_orig.data.to(p.device)
    keep = np.float64 if spec.dtype == torch.float64 else np.float32
    out = logits.squeeze(0).numpy().astype(keep)
    if also_online and not online:
        out_online = stitch_ref(online_outputs, torch.zeros_like(all_logits), torch.zeros_like(logit_count)).squeeze(0).numpy().astype(keep)
        return (out, out_online, updated) if return_params else (out, out_online)
    return (out, updated) if return_params else out
