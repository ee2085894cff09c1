"""Drop-in mirror of the reference's dynamic-eval library for the hot path (reference lcasr/lib.py).

Same names, argument meaning and error behaviour as the reference for: `prepare_chunks` (lib.py:128-145),
`get_specaugment_config_from_args` (:102-112), `get_frame_shuffle_config_from_args` (:114-120),
`get_lr_args_from_args` (:122-125), `get_cutout_params_from_args` (:419-428), the freeze helpers (:163-204),
`dynamic_eval_ctc_loss` / `dynamic_eval` (:450-643) and `apply_args` (:1756-1787).

What changes is WHERE the work runs (MI355X-first):
  * the recording's log-mel is uploaded once and windows are views of it in HBM (the reference does one H2D per
    window, lib.py:549);
  * posteriors never leave the device: greedy decode, CTC loss + gradient, stitching (exp / overlap-add / log) are
    HIP kernels; only token ids (a few hundred int32) cross PCIe per window (reference: 2 D2H of [T/8, V+1] per
    step + one per window in the final pass, lib.py:559,565,604);
  * weight snapshot / restore is one device-to-device copy of the flat parameter buffer (reference: through host
    memory, lib.py:482-483,636-637);
  * forward/backward of the acoustic model and the MADGRAD/Adam step are our kernels (model.py, optim.py).
There is no CPU fallback anywhere on this path."""
import random
import time

import torch

from . import ops
from . import augment as _aug
from .augment import SpecAugment
from .decoding import GreedyCTCDecoder
from .optim import MADGRAD, Adam  # noqa: F401  (re-exported: `optim=lib.MADGRAD`)

try:  # tqdm is optional plumbing
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, **_):
        return x


# ------------------------------------------------------------------------------------------------ arg -> config helpers
def get_specaugment_config_from_args(args):
    a = {k.replace('spec_augment_', ''): v for k, v in args.__dict__.items() if k.startswith('spec_augment')}
    return {
        'n_time_masks': a.get('n_time_masks', 0),
        'n_freq_masks': a.get('n_freq_masks', 0),
        'freq_mask_param': a.get('freq_mask_param', 42),
        'time_mask_param': a.get('time_mask_param', -1),
        'min_p': a.get('min_p', 0.05),
        'zero_masking': a.get('zero_masking', False),
    }


def get_frame_shuffle_config_from_args(args):
    a = {k.replace('frame_shuffle_', ''): v for k, v in args.__dict__.items() if k.startswith('frame_shuffle')}
    return {'time_dimension': a.get('time_dimension', False), 'freq_dimension': a.get('freq_dimension', False)}


def get_lr_args_from_args(args):
    lr_args = {k.replace('optim_', ''): v for k, v in args.__dict__.items() if k.startswith('optim_')}
    lr_args['lr'] = lr_args.get('lr', 9e-5)
    return lr_args


def get_cutout_params_from_args(args, seq_len):
    a = {k.replace('cutout_', ''): v for k, v in args.__dict__.items() if k.startswith('cutout')}
    return {
        'seq_len': seq_len,
        'cutout_val': a.get('value', 'mean'),
        'num_rectangles': a.get('num_rectangles', 0),
        'max_width': a.get('max_width', 100),
        'max_height': a.get('max_height', 10),
    }


def prepare_chunks(spec, seq_len, overlap):
    """Window index set, exactly as reference lcasr/lib.py:128-145: stride seq_len - overlap; windows are kept until
    ONE window shorter than its predecessor has been added; a single window when the recording fits."""
    spec_n = spec.shape[-1]
    last_ulen, kill_next = None, False
    if spec_n <= seq_len:
        return {0: spec}, [0]
    training_data = {}
    for i in range(0, spec_n, seq_len - overlap):
        audio_chunk = spec[:, :, i:i + seq_len]  # [B, C, T] view
        u_len = audio_chunk.shape[-1]
        if kill_next:
            break
        elif last_ulen is not None and u_len < last_ulen:
            kill_next = True
        last_ulen = u_len
        training_data[i] = audio_chunk
    return training_data, list(training_data.keys())


# ------------------------------------------------------------------------------------------------ freeze helpers
def _set_frozen(model, prefixes=None, only=None):
    if hasattr(model, "frozen"):  # our SCConformerXL
        if only is not None:
            model.frozen = {n for n, _ in model.spec if not any(n.startswith(o) for o in only)}
        else:
            model.frozen = set(model.frozen) | set(prefixes)
    else:  # foreign torch module: same attribute the reference flips
        for n, p in model.named_parameters():
            if only is not None:
                p.requires_grad = any(n.startswith(o) for o in only)
            elif any(n.startswith(f) for f in prefixes):
                p.requires_grad = False
    return model


def bitfit(model):
    """reference lcasr/lib.py:148-160: freeze everything, then train only the biases of LayerNorm / Linear / BatchRenorm modules.
    Upstream module classes are not visible (un-vendored), so the mapping onto this package's parameter names is: every
    `*norm*.bias` (block, module and decoder LayerNorms, the conv-module norm when it has a bias) and the biases of the Linear
    layers (`attn.qkv`, `attn.out`, `subsampling.out`, `decoder.ff`, `decoder.reproj`); convolution biases (depthwise and
    pointwise convs are Conv modules upstream, not nn.Linear) stay frozen like the weights."""
    if not hasattr(model, "frozen"):
        raise ops.DynError("bitfit: only the HIP SCConformerXL is supported")
    linear_bias = ("attn.qkv.bias", "attn.out.bias", "subsampling.out.bias", "decoder.ff.bias", "decoder.reproj.bias")
    train = {n for n, _ in model.spec if n.endswith(".bias") and ("norm" in n.rsplit(".", 2)[-2] or n.endswith(linear_bias))}
    model.frozen = {n for n, _ in model.spec if n not in train}
    return model


def freeze_subsampling(model):
    if getattr(model, 'subsampling', None) is None:
        print('No subsampling module found to freeze')
        return model
    print('Freezing subsampling module')
    return _set_frozen(model, prefixes=['subsampling.'])


def freeze_all_but_last_block_and_head(model):
    n = len(model.layers)
    print(f'Training only last block: layers.{n - 1} and CTC head')
    return _set_frozen(model, only=[f'layers.{n - 1}.', 'decoder.'])


def train_subsampling_only(model):
    if getattr(model, 'subsampling', None) is None:
        print('No subsampling module found to train')
        return model
    print('Training only subsampling module')
    return _set_frozen(model, only=['subsampling.'])


# ------------------------------------------------------------------------------------------------ the hot path
def _is_native(model):
    return hasattr(model, "flat_params") and hasattr(model, "backward")


def _window_fill_value(window, zero_masking):
    """Fill value of the SpecAugment masks: 0, or the window mean as a 1-element DEVICE tensor (dyn_moments) so that it
    never makes a host round trip."""
    if zero_masking:
        return 0.0
    from ._lib import check, load
    out = torch.empty(3, device=window.device, dtype=torch.float32)
    ws = ops.workspace(window.device)
    check(load().dyn_moments(window.data_ptr(), window.numel(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                             torch.cuda.current_stream().cuda_stream), "dyn_moments")
    return out[1:2]


def _unsupported(name):
    raise NotImplementedError(f"{name} is an optional augmentation of the reference that the HIP path does not implement "
                              "yet; refusing to silently run without it")


def entropy_augmentation(spec, model, **kwargs):
    """Reference lcasr/lib.py:86-99: spec += 0.001 * d(mean entropy of the posteriors)/d(spec), in place on the device
    window(s) `spec` [B, F, T].  Forward, entropy gradient (dyn_entropy_grad), backward to the input with every
    weight-gradient product skipped."""
    if not kwargs.get('enabled', False):
        return spec
    with torch.enable_grad():
        lp = model(audio_signal=spec)['final_posteriors']
    g, _ = ops.entropy_grad(lp, 1.0 / (lp.shape[0] * lp.shape[1]))        # entropy.mean() over B * N rows
    dx = model.backward(g, input_grad=True, param_grads=False)
    for b in range(spec.shape[0]):
        ops.axpby(dx[b], spec[b], a=0.001, b=1.0)
    return spec


def _dynamic_eval_gen(
        args,
        model,
        spec: torch.Tensor,
        seq_len: int,
        overlap: int,
        tokenizer,
        use_tqdm=True,
        optim=MADGRAD,
        optimizer_state: dict = None,
        beam_search_fn=None,
        return_params: bool = False,
        return_device: bool = False,
):
    """Generator form of dynamic_eval_ctc_loss: yields where the host would otherwise block on the GPU (the per-window
    pseudo-label ids) or has queued a batch of independent work, so a driver can interleave several recording chains on
    separate streams from ONE host thread (dynamic_eval_many).  The return value travels in StopIteration.value."""
    if beam_search_fn is not None and args.__dict__.get('lm_tta_beams', 3) != 0:
        _unsupported("LM beam-search pseudo-labels (beam_search_fn)")
    device = model.device
    if torch.device(device).type != "cuda":
        raise ops.DynError("dynamic_eval: model.device must be a GPU (no CPU fallback)")
    spec_n = spec.shape[-1]
    downsampling_factor = args.config['model']['subsampling_factor']
    seq_len = seq_len if seq_len != -1 else args.config['audio_chunking']['size']

    spec_augment_config = get_specaugment_config_from_args(args)
    random_noise = args.__dict__.get('random_noise', 0.0)
    lr_args = get_lr_args_from_args(args)
    frame_shuffle_args = get_frame_shuffle_config_from_args(args)
    entropy_args = {k.replace('entropy_augmentation_', ''): v for k, v in args.__dict__.items()
                    if k.startswith('entropy_augmentation_')}
    cutout_args = get_cutout_params_from_args(args, seq_len)
    verbose = bool(args.__dict__.get('verbose', False)) and not args.__dict__.get('quiet', False)
    if verbose:
        print(spec_augment_config, lr_args, frame_shuffle_args, cutout_args)
    if entropy_args.get('enabled', False) and not _is_native(model):
        _unsupported("entropy_augmentation with a foreign torch model")
    num_negatives = 1
    native = _is_native(model)

    # snapshot of the weights, kept in HBM (reference: CPU clones, lib.py:482-483)
    if native:
        original_flat = model.flat_params.clone()
        frozen_before = set(model.frozen)
    else:
        original_model_params = [p.clone().detach() for p in model.parameters()]

    if args.__dict__.get('freeze_subsampling', False):
        model = freeze_subsampling(model)
    if args.__dict__.get('freeze_all_but_last_block_and_head', False):
        model = freeze_all_but_last_block_and_head(model)
    if args.__dict__.get('train_subsampling_only', False):
        model = train_subsampling_only(model)

    num_classes = model.decoder.num_classes
    blank = num_classes - 1
    optimizer = optim(model.parameters(), **lr_args)
    if optimizer_state is not None:
        optimizer.load_state_dict(optimizer_state)

    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=blank, device=device)
    augmentation = SpecAugment(**spec_augment_config)
    fixed_masks = args.__dict__.get('spec_augment_fixed_masks', None)  # test hook: {window_key: masks}

    if seq_len > spec_n:
        seq_len, overlap = spec_n, 0
    else:
        overlap = overlap if overlap != -1 else args.config['audio_chunking']['overlap']

    assert args.config['training'].get("max_seq_len", 0) == 0, 'caching is not used anymore'
    assert overlap / downsampling_factor == overlap // downsampling_factor, 'Overlap must be a multiple of the downsampling factor'
    if verbose:
        print(f'Using seq_len: {seq_len} and overlap: {overlap}')
    assert tokenizer.vocab_size() + 1 == num_classes, 'tokenizer vocabulary does not match the CTC head'

    epochs = args.__dict__.get('epochs', 1)
    shuffle = args.__dict__.get('shuffle', False)
    online = args.__dict__.get('online', False)
    epochs = 1 if online else epochs
    shuffle = False if online else shuffle
    print_runtimes = args.__dict__.get('print_runtimes', False)
    skip_zero = args.__dict__.get('skip_zero_grad_samples', True)
    final_batch = int(args.__dict__.get('final_pass_batch', 4))
    if print_runtimes:
        print('Spectrogram length:', spec_n)

    # the whole recording lives in HBM; windows are views (one upload instead of one per window)
    spec_dev = spec.to(device=device, dtype=torch.float32)
    if spec_dev.dim() != 3 or spec_dev.shape[0] != 1:
        raise ops.DynError(f"spec must be [1, F, T], got {tuple(spec.shape)}")
    Fq = spec_dev.shape[1]

    # on-device stitch accumulators (reference: two host buffers of spec_n//4 + seq_len rows, lib.py:510)
    acc_rows = spec_n // 4 + seq_len
    acc = torch.zeros(acc_rows, num_classes, device=device, dtype=torch.float32)
    cnt = torch.zeros(acc_rows, device=device, dtype=torch.float32)
    stitch = {"pos": 0, "end": 0}

    def stitch_window(key, log_probs_2d, u_len):
        ds_len = log_probs_2d.shape[0]
        ratio = u_len / ds_len
        overlap_ds = int(overlap / ratio)
        stitch["pos"] -= overlap_ds if key != 0 else 0
        ops.stitch_accumulate(log_probs_2d, acc, cnt, stitch["pos"])
        stitch["pos"] += ds_len
        stitch["end"] = max(stitch["end"], stitch["pos"])

    pinned = None
    tgt_ring, tgt_turn = None, 0
    if native:
        model.use_graphs = bool(args.__dict__.get('use_graphs', True))   # hipGraph replay of the per-window launch sequences
        # only the augmented copies are differentiated (lib.py:570-575): the clean copy's attention need not keep its probabilities
        model.grad_samples = num_negatives if (skip_zero and _CLEAN_COPY_FUSED_ATTN) else None
    model.eval()  # don't update batchrenorm (reference lib.py:525)
    training_data, training_keys = prepare_chunks(spec_dev, seq_len, overlap)
    for epoch in range(args.__dict__.get('epochs', 1)):
        if verbose:
            print(f'Epoch {epoch + 1} / {epochs}')
        if online and epoch > 0:
            # the reference's loop runs range(args.epochs) even in online mode (lib.py:527) and every epoch overwrites
            # model_outputs[i] (lib.py:589): only the last epoch's posteriors are stitched
            acc.zero_(); cnt.zero_()
            stitch["pos"] = stitch["end"] = 0
        training_keys = list(training_data.keys())
        training_keys = random.sample(training_keys, len(training_keys)) if shuffle else training_keys
        epochs_stime = time.time()
        pbar = tqdm(training_keys) if use_tqdm else training_keys
        for i in pbar:
            sampled = 0
            if ops.gemm_profile_active():        # bench.py's live roofline sampling; everything it needs in the loop is in these blocks
                sampled = ops.gemm_profile_begin_step(device)
            view = training_data[i][0]  # [F, u_len] view into the recording
            u_len = view.shape[-1]
            audio_chunk = torch.empty(num_negatives + 1, Fq, u_len, device=device, dtype=torch.float32)
            for b in range(num_negatives + 1):
                audio_chunk[b].copy_(view)
            for b in range(num_negatives):  # augment copy 0, copy -1 stays clean (reference lib.py:541)
                masks = fixed_masks[i] if fixed_masks is not None else augmentation.draw(Fq, u_len)
                if masks[0][0] or masks[1][0]:
                    fill = _window_fill_value(audio_chunk[b], augmentation.zero_masking)
                    augmentation.apply(audio_chunk[b], masks, fill)
                _aug.frame_shuffle(audio_chunk[b], **frame_shuffle_args)                 # reference lib.py:542
                _aug.add_random_noise(audio_chunk[b], noise_factor=random_noise)         # lib.py:543
                _aug.cutout(audio_chunk[b], **cutout_args)                               # lib.py:544
            if entropy_args.get('enabled', False):
                entropy_augmentation(audio_chunk[:num_negatives], model, **entropy_args)  # lib.py:545

            with torch.enable_grad():
                out = model(audio_signal=audio_chunk)
            post = out['final_posteriors']  # [B, N, C] on device

            # greedy ids on device (reference lib.py:559); only the ids cross PCIe, asynchronously into pinned memory
            ids_dev, n_dev = ops.ctc_greedy(post[-1].detach(), blank)
            if pinned is None or pinned[0].shape[1] < ids_dev.shape[1]:
                pinned = (torch.empty(1, ids_dev.shape[1], dtype=torch.int32, pin_memory=True), torch.empty(1, dtype=torch.int32, pin_memory=True))
            pinned[0][:, :ids_dev.shape[1]].copy_(ids_dev, non_blocking=True)
            pinned[1].copy_(n_dev, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
            if sampled:
                ops.gemm_profile_before_yield(sampled, ready)
            yield                                            # another chain may use the host while this forward runs
            _t0 = time.perf_counter()
            if ops.gemm_profile_active():        # also for unsampled steps: the mode is per model call, chains interleave on this thread
                ops.gemm_profile_resume_step(device, sampled)
            ready.synchronize()
            HOST_WAIT[0] += time.perf_counter() - _t0
            pseudo_targets = tokenizer.decode(pinned[0][0, :int(pinned[1][0])].tolist())
            if verbose and not args.__dict__.get('not_verbose', False) and args.__dict__.get('print_predictions', False):
                print(f'Pseudo targets: {pseudo_targets}')
                print(f'Noisy predictions: {decoder(post[0].detach())}\n--\n')
            target_ids = tokenizer.encode(pseudo_targets)  # text hop kept (reference lib.py:569)
            S = len(target_ids)
            # pseudo-label ids go up through a small ring of pinned buffers with an async copy (a pageable upload would
            # block the host on this stream); a slot is reused 4 windows later, long after its copy has run
            if tgt_ring is None or tgt_ring[0].shape[1] < max(S, 1):
                tgt_ring = [torch.empty(num_negatives, max(2 * S, 256), dtype=torch.int32, pin_memory=True) for _ in range(4)]
            slot = tgt_ring[tgt_turn % 4]
            tgt_turn += 1
            row = torch.as_tensor(target_ids if S else [0], dtype=torch.int32)
            slot[:, :row.numel()] = row
            targets = torch.empty(num_negatives, max(S, 1), dtype=torch.int32, device=device)
            targets.copy_(slot[:, :max(S, 1)], non_blocking=True)
            augmented_outs = post[:num_negatives]
            N, B = augmented_outs.shape[1], augmented_outs.shape[0]
            total_tokens_in_loss = N * B
            ilen = torch.full((B,), N, dtype=torch.int32, device=device)
            tlen = torch.full((B,), S, dtype=torch.int32, device=device)

            if native:
                # CTCLoss(reduction='sum') / (N*B) and its gradient w.r.t. the log-probs (reference lib.py:575,579)
                _, _, g_aug = ops.ctc_loss(augmented_outs.contiguous(), targets, ilen, tlen, blank, reduction="sum",
                                           grad_scale=1.0 / total_tokens_in_loss)
                optimizer.zero_grad()
                if skip_zero:
                    model.backward(g_aug, n_active=num_negatives)  # the clean copy's gradient is identically zero
                else:
                    g_full = torch.zeros_like(post)
                    g_full[:num_negatives].copy_(g_aug)
                    model.backward(g_full)
            else:
                _, _, g_aug = ops.ctc_loss(augmented_outs.detach().contiguous(), targets, ilen, tlen, blank, reduction="sum",
                                           grad_scale=1.0 / total_tokens_in_loss)
                optimizer.zero_grad()
                augmented_outs.backward(g_aug)
            optimizer.step()

            if online:
                stitch_window(i, post[-1].detach(), u_len)
            if sampled:
                ops.gemm_profile_end_step(device, sampled)
        epochs_etime = time.time()
        if print_runtimes:
            torch.cuda.synchronize(device)
            print(f'Epoch runtime: {time.time() - epochs_stime}')
            del epochs_etime

    if not online:
        model.eval()
        training_data, training_keys = prepare_chunks(spec_dev, seq_len, overlap)
        final_pass_stime = time.time()
        keys = sorted(training_keys)
        idx = 0
        while idx < len(keys):
            # windows are independent here: batch equal-length ones to fill the GPU (reference: B = 1, lib.py:599-609)
            group = [keys[idx]]
            u_len = training_data[keys[idx]].shape[-1]
            while len(group) < final_batch and idx + len(group) < len(keys) and \
                    training_data[keys[idx + len(group)]].shape[-1] == u_len:
                group.append(keys[idx + len(group)])
            sampled = 0
            if ops.gemm_profile_active():
                sampled = ops.gemm_profile_begin_step(device)
            batch = torch.empty(len(group), Fq, u_len, device=device, dtype=torch.float32)
            for b, k in enumerate(group):
                batch[b].copy_(training_data[k][0])
            with torch.no_grad():   # never held across a yield: interleaved generators would restore each other's grad mode
                post = model(audio_signal=batch)['final_posteriors']
                for b, k in enumerate(group):
                    stitch_window(k, post[b], u_len)
            idx += len(group)
            if sampled:
                ops.gemm_profile_end_step(device, sampled)
            yield                                        # independent forwards are queued: let another chain enqueue
        if print_runtimes:
            torch.cuda.synchronize(device)
            print(f'Final pass runtime: {time.time() - final_pass_stime}')
        model.train()

    logits_dev = ops.stitch_finalize(acc, cnt, stitch["end"])  # log(sum / count) over the covered rows

    if return_params:
        updated_model_params = [p.clone().detach().cpu() for p in model.parameters()]

    # reset model parameters (reference lib.py:636-637)
    if native:
        model.flat_params.copy_(original_flat)
        model.frozen = frozen_before
        model.grad_samples = None
    else:
        for p, p_orig in zip(model.parameters(), original_model_params):
            p.data = p_orig.data.to(p.device)

    logits = logits_dev if return_device else logits_dev.cpu().numpy()
    return logits if not return_params else (logits, updated_model_params)


def lockstep_supported(args, model, specs, beam_search_fn=None, optimizer_state=None):
    """Can `specs` go through ONE lockstep group of `model` (SCConformerXL(group=R))?  The group form covers the standard recipe (SpecAugment
    on copy 0, MADGRAD / Adam, online or final pass, epochs, shuffle, recordings of different lengths); the optional augmentations,
    optimiser state hand-over and LM beam search keep the one-recording-per-chain path."""
    a = args.__dict__
    if not (_is_native(model) and getattr(model, "R", 1) > 1 and 1 <= len(specs) <= model.R):
        return False
    if beam_search_fn is not None or optimizer_state is not None:
        return False
    if a.get('random_noise', 0.0) or any(a.get(k, False) for k in ('freeze_subsampling', 'freeze_all_but_last_block_and_head', 'train_subsampling_only')):
        return False
    if any(v for v in get_frame_shuffle_config_from_args(args).values()) or a.get('cutout_num_rectangles', 0) or a.get('entropy_augmentation_enabled', False):
        return False
    if not a.get('skip_zero_grad_samples', True):
        return False
    return len({int(sp.shape[-2]) for sp in specs}) == 1


def _shape_classes(members, u_lens):
    """Contiguous runs [lo, hi) of `members` (ascending replica indices) whose windows have the same length."""
    out, k = [], 0
    while k < len(members):
        j = k
        while j + 1 < len(members) and members[j + 1] == members[j] + 1 and u_lens[members[j + 1]] == u_lens[members[k]]:
            j += 1
        out.append((members[k], members[j] + 1))
        k = j + 1
    return out


def _dynamic_eval_group_gen(args, model, specs, seq_len, overlap, tokenizer, use_tqdm=True, optim=MADGRAD, optimizer_state=None,
                            beam_search_fn=None, return_params=False, return_device=False):
    """dynamic_eval_ctc_loss (reference lcasr/lib.py:450-640) for R' = len(specs) recordings in lockstep on one SCConformerXL(group=R >= R'):
    every window step runs ONCE for all recordings that have a window of that shape at that position — batch [2 n, F, T] ordered (augmented
    copies of the n recordings, then their clean copies), one forward, one greedy decode, one CTC launch over the n augmented copies (each
    recording's loss and gradient scaled as if it were alone: reduction 'sum', 1 / (N * B) with B = 1), one backward on the n augmented
    samples — and ONE optimiser launch over the [n, n_flat] buffers of the recordings still adapting.  Recordings of different lengths share
    the window grid (same seq_len / overlap): they are ordered longest first, the full windows run on a shrinking prefix of the replicas,
    a recording's short last window runs on its own replica, a finished recording's replica is no longer stepped, and every recording keeps
    its own optimiser step count (over several epochs, or under shuffle, recordings of different lengths get out of step).  Recordings stay
    independent (own weights, own optimiser state, own stitch buffers): per recording the results are those of `dynamic_eval` up to the GEMM
    planner's choice of tile for the larger launches.  Generator with the same yield points as _dynamic_eval_gen; returns the list of
    per-recording results (in `specs` order) in StopIteration.value."""
    if not lockstep_supported(args, model, specs, beam_search_fn, optimizer_state):
        raise ops.DynError("dynamic_eval lockstep group: unsupported configuration (see lockstep_supported)")
    device = model.device
    Rn = len(specs)
    order = sorted(range(Rn), key=lambda j: (-int(specs[j].shape[-1]), j))        # replica q holds recording order[q]: longest first
    downsampling_factor = args.config['model']['subsampling_factor']
    seq_len = seq_len if seq_len != -1 else args.config['audio_chunking']['size']
    spec_augment_config = get_specaugment_config_from_args(args)
    lr_args = get_lr_args_from_args(args)
    num_negatives = 1
    prev_range = (model._lo, model._lo + model._n)
    model.set_range(0, Rn)
    n_flat = model.n_flat
    original_flat = model.flat_params[:Rn * n_flat].clone()
    num_classes = model.decoder.num_classes
    blank = num_classes - 1
    optimizer = optim(model.parameters(), **lr_args)             # flat [R', n_flat] buffers: one launch per step for the whole group
    augmentation = SpecAugment(**spec_augment_config)
    fixed_masks = args.__dict__.get('spec_augment_fixed_masks', None)  # test hook: {window_key: masks} or one such dict per recording
    assert args.config['training'].get("max_seq_len", 0) == 0, 'caching is not used anymore'
    assert tokenizer.vocab_size() + 1 == num_classes, 'tokenizer vocabulary does not match the CTC head'
    epochs = args.__dict__.get('epochs', 1)
    shuffle = args.__dict__.get('shuffle', False)
    online = args.__dict__.get('online', False)
    shuffle = False if online else shuffle
    final_batch = max(1, int(args.__dict__.get('final_pass_batch', 4)))
    specs_dev = []
    for q in range(Rn):
        sp = specs[order[q]].to(device=device, dtype=torch.float32)
        if sp.dim() != 3 or sp.shape[0] != 1:
            raise ops.DynError(f"spec must be [1, F, T], got {tuple(sp.shape)}")
        specs_dev.append(sp)
    Fq = specs_dev[0].shape[1]
    # per recording: window rule and accumulators exactly as the single path (a recording shorter than seq_len is one window, overlap 0)
    seqs, ovls, data, acc, cnt, stitch = [], [], [], [], [], []
    for q in range(Rn):
        spec_n = specs_dev[q].shape[-1]
        sl, ov = (spec_n, 0) if seq_len > spec_n else (seq_len, overlap if overlap != -1 else args.config['audio_chunking']['overlap'])
        assert ov / downsampling_factor == ov // downsampling_factor, 'Overlap must be a multiple of the downsampling factor'
        seqs.append(sl); ovls.append(ov)
        data.append(prepare_chunks(specs_dev[q], sl, ov)[0])
        rows = spec_n // 4 + sl
        acc.append(torch.zeros(rows, num_classes, device=device, dtype=torch.float32))
        cnt.append(torch.zeros(rows, device=device, dtype=torch.float32))
        stitch.append({"pos": 0, "end": 0})
    if len({(sl - ov) for sl, ov, d in zip(seqs, ovls, data) if len(d) > 1}) > 1:
        raise ops.DynError("lockstep group: recordings must share the window stride")

    def stitch_window(q, key, log_probs_2d, u_len):
        ds_len = log_probs_2d.shape[0]
        overlap_ds = int(ovls[q] / (u_len / ds_len))
        st = stitch[q]
        st["pos"] -= overlap_ds if key != 0 else 0
        ops.stitch_accumulate(log_probs_2d, acc[q], cnt[q], st["pos"])
        st["pos"] += ds_len
        st["end"] = max(st["end"], st["pos"])

    model.use_graphs = bool(args.__dict__.get('use_graphs', True))
    model.eval()
    all_keys = sorted(set().union(*[set(d.keys()) for d in data]))
    ksteps = [0] * Rn            # every recording keeps its own optimiser step count (recordings of different lengths get out of step after one epoch)
    pinned = None
    tgt_ring, tgt_turn = None, 0
    results = [None] * Rn
    try:
        for epoch in range(args.__dict__.get('epochs', 1)):
            if online and epoch > 0:
                for q in range(Rn):
                    acc[q].zero_(); cnt[q].zero_()
                    stitch[q]["pos"] = stitch[q]["end"] = 0
            training_keys = random.sample(all_keys, len(all_keys)) if shuffle else list(all_keys)
            for i in (tqdm(training_keys) if use_tqdm else training_keys):
                members = [q for q in range(Rn) if i in data[q]]
                u_lens = {q: data[q][i].shape[-1] for q in members}
                classes = _shape_classes(members, u_lens)
                sampled = 0
                if ops.gemm_profile_active():
                    sampled = ops.gemm_profile_begin_step(device)
                posts = []
                for lo, hi in classes:               # forward of every shape class; the labels of all of them cross PCIe together
                    n = hi - lo
                    u_len = u_lens[lo]
                    audio_chunk = torch.empty(2 * n, Fq, u_len, device=device, dtype=torch.float32)
                    for q in range(lo, hi):
                        view = data[q][i][0]
                        audio_chunk[q - lo].copy_(view)
                        audio_chunk[n + q - lo].copy_(view)
                        fm = fixed_masks[order[q]] if isinstance(fixed_masks, (list, tuple)) else fixed_masks
                        masks = fm[i] if fm is not None else augmentation.draw(Fq, u_len)
                        if masks[0][0] or masks[1][0]:
                            augmentation.apply(audio_chunk[q - lo], masks, _window_fill_value(audio_chunk[q - lo], augmentation.zero_masking))
                    model.set_range(lo, hi)
                    model.grad_samples = num_negatives * n if _CLEAN_COPY_FUSED_ATTN else None
                    with torch.enable_grad():
                        post = model(audio_signal=audio_chunk)['final_posteriors']     # [2 n, N, C]
                    ctx = (model._ctx, model._ctx_static, model._ctx_key)
                    ids_dev, n_dev = ops.ctc_greedy(post[n:].detach(), blank)       # pseudo-labels of the clean copies
                    if pinned is None or pinned[0].shape[1] < ids_dev.shape[1]:
                        pinned = (torch.empty(Rn, max(ids_dev.shape[1], seq_len // downsampling_factor), dtype=torch.int32, pin_memory=True),
                                  torch.empty(Rn, dtype=torch.int32, pin_memory=True))
                    pinned[0][lo:hi, :ids_dev.shape[1]].copy_(ids_dev, non_blocking=True)
                    pinned[1][lo:hi].copy_(n_dev, non_blocking=True)
                    posts.append((post, ctx))
                ready = torch.cuda.Event()
                ready.record()
                if sampled:
                    ops.gemm_profile_before_yield(sampled, ready)
                yield
                _t0 = time.perf_counter()
                if ops.gemm_profile_active():
                    ops.gemm_profile_resume_step(device, sampled)
                ready.synchronize()
                HOST_WAIT[0] += time.perf_counter() - _t0
                optimizer.zero_grad()
                for (lo, hi), (post, ctx) in zip(classes, posts):
                    n = hi - lo
                    target_ids = []
                    for q in range(lo, hi):
                        pseudo_targets = tokenizer.decode(pinned[0][q, :int(pinned[1][q])].tolist())
                        target_ids.append(tokenizer.encode(pseudo_targets))                   # text hop kept (reference lib.py:569)
                    S_max = max(1, max(len(t) for t in target_ids))
                    if tgt_ring is None or tgt_ring[0][0].shape[1] < S_max:
                        tgt_ring = [(torch.zeros(Rn, max(2 * S_max, 256), dtype=torch.int32, pin_memory=True),
                                     torch.zeros(Rn, dtype=torch.int32, pin_memory=True)) for _ in range(4 * max(1, len(classes)))]
                    slot, lens = tgt_ring[tgt_turn % len(tgt_ring)]
                    tgt_turn += 1
                    for k, t in enumerate(target_ids):
                        if t:
                            slot[k, :len(t)] = torch.as_tensor(t, dtype=torch.int32)
                        lens[k] = len(t)
                    targets = torch.empty(n, S_max, dtype=torch.int32, device=device)
                    targets.copy_(slot[:n, :S_max], non_blocking=True)
                    tlen = torch.empty(n, dtype=torch.int32, device=device)
                    tlen.copy_(lens[:n], non_blocking=True)
                    N = post.shape[1]
                    ilen = torch.full((n,), N, dtype=torch.int32, device=device)
                    # per recording: CTCLoss(reduction='sum') / (N * B) with B = num_negatives = 1 (reference lib.py:572-575); 'sum' over the
                    # class's samples leaves every sample its own gradient
                    _, _, g_aug = ops.ctc_loss(post[:n].contiguous(), targets, ilen, tlen, blank, reduction="sum", grad_scale=1.0 / (N * num_negatives))
                    model.set_range(lo, hi)
                    model._ctx, model._ctx_static, model._ctx_key = ctx
                    model.backward(g_aug, n_active=n)
                    if online:
                        for q in range(lo, hi):
                            stitch_window(q, i, post[n + q - lo].detach(), u_lens[q])
                # one optimiser launch per run of recordings with the same step count (one launch when they are in step), over the recordings that
                # had a window at this position; a finished recording's replica is left alone
                runs, start, prev = [], members[0], members[0]
                for q in members[1:]:
                    if q == prev + 1 and ksteps[q] == ksteps[start]:
                        prev = q
                        continue
                    runs.append((start * n_flat, (prev + 1) * n_flat, ksteps[start]))
                    start = prev = q
                runs.append((start * n_flat, (prev + 1) * n_flat, ksteps[start]))
                optimizer.step_ranges(runs)
                for q in members:
                    ksteps[q] += 1
                if sampled:
                    ops.gemm_profile_end_step(device, sampled)
        if not online:
            model.eval()
            idx = 0
            while idx < len(all_keys):
                k0 = all_keys[idx]
                members = [q for q in range(Rn) if k0 in data[q]]
                u_lens = {q: data[q][k0].shape[-1] for q in members}
                classes = _shape_classes(members, u_lens)
                lo, hi = classes[0]                     # the class of the longest recordings: batch consecutive positions of the same class
                group = [k0]
                while len(group) < final_batch and idx + len(group) < len(all_keys):
                    kn = all_keys[idx + len(group)]
                    mem_n = [q for q in range(Rn) if kn in data[q]]
                    ul_n = {q: data[q][kn].shape[-1] for q in mem_n}
                    if not mem_n or _shape_classes(mem_n, ul_n)[0] != (lo, hi) or ul_n[lo] != u_lens[lo]:
                        break
                    group.append(kn)
                sampled = 0
                if ops.gemm_profile_active():
                    sampled = ops.gemm_profile_begin_step(device)
                n = hi - lo
                batch = torch.empty(len(group) * n, Fq, u_lens[lo], device=device, dtype=torch.float32)
                for c, k in enumerate(group):
                    for q in range(lo, hi):
                        batch[c * n + q - lo].copy_(data[q][k][0])
                model.set_range(lo, hi)
                with torch.no_grad():
                    post = model(audio_signal=batch)['final_posteriors']
                    for c, k in enumerate(group):
                        for q in range(lo, hi):
                            stitch_window(q, k, post[c * n + q - lo], u_lens[lo])
                    for k in group:                 # the other classes at these positions (short last windows): one forward each
                        mem_k = [q for q in range(Rn) if k in data[q] and not (lo <= q < hi)]
                        ul_k = {q: data[q][k].shape[-1] for q in mem_k}
                        for l2, h2 in _shape_classes(mem_k, ul_k):
                            b2 = torch.empty(h2 - l2, Fq, ul_k[l2], device=device, dtype=torch.float32)
                            for q in range(l2, h2):
                                b2[q - l2].copy_(data[q][k][0])
                            model.set_range(l2, h2)
                            p2 = model(audio_signal=b2)['final_posteriors']
                            for q in range(l2, h2):
                                stitch_window(q, k, p2[q - l2], ul_k[l2])
                idx += len(group)
                if sampled:
                    ops.gemm_profile_end_step(device, sampled)
                yield
            model.train()
        for q in range(Rn):
            logits_dev = ops.stitch_finalize(acc[q], cnt[q], stitch[q]["end"])
            logits = logits_dev if return_device else logits_dev.cpu().numpy()
            if return_params:
                results[order[q]] = (logits, [p.clone().detach().cpu() for p in model.replica_params(q)])
            else:
                results[order[q]] = logits
    finally:
        model.flat_params[:Rn * n_flat].copy_(original_flat)      # reference lib.py:636-637
        model.grad_samples = None
        model.set_range(*prev_range)
    return results


def dynamic_eval_lockstep(args, model, specs, seq_len, overlap, tokenizer, **kw):
    """`specs` (<= model.R recordings of equal length) through one lockstep group; list of per-recording results (see _dynamic_eval_group_gen)."""
    gen = _dynamic_eval_group_gen(args, model, specs, seq_len, overlap, tokenizer, **kw)
    try:
        while True:
            next(gen)
    except StopIteration as stop:
        return stop.value


def dynamic_eval_ctc_loss(args, model, spec, seq_len, overlap, tokenizer, use_tqdm=True, optim=MADGRAD, optimizer_state=None,
                          beam_search_fn=None, return_params=False, return_device=False):
    """Reference lcasr/lib.py:450-640.  Returns np.float32 [T_ds, V+1] log-probs (and the adapted parameters as CPU
    clones when `return_params`).  `return_device=True` (extension) returns the stitched log-probs as a CUDA tensor so
    the harness can decode them without a PCIe round trip."""
    gen = _dynamic_eval_gen(args, model, spec, seq_len, overlap, tokenizer, use_tqdm=use_tqdm, optim=optim,
                            optimizer_state=optimizer_state, beam_search_fn=beam_search_fn, return_params=return_params,
                            return_device=return_device)
    try:
        while True:
            next(gen)
    except StopIteration as stop:
        return stop.value
    finally:
        if _is_native(model):
            model.grad_samples = None      # also when the loop raised: a stale value would make a later full-batch backward fail


def _new_chain_stream(device, k):
    """Stream of recording chain k.  DYN_CHAIN_CU_MASK=<n>[:stride] (experiment switch, off by default) gives chain k a stream whose
    kernels may not use a group of n of the 256 CUs — CUs k*n .. k*n+n-1, or with `:stride` every (256/n)-th CU starting at k — so the
    short kernels of the OTHER chains can start there while a matrix kernel of chain k holds the rest of the chip
    (dyn_stream_create_cu_mask = hipExtStreamCreateWithCUMask).  Measured: DESIGN.md §5."""
    import ctypes
    import os
    spec = os.environ.get("DYN_CHAIN_CU_MASK", "")
    if not spec or spec == "0":
        return torch.cuda.Stream(device=device)
    from ._lib import check, load
    n = int(spec.split(":")[0])
    strided = spec.endswith(":stride")
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    if not 0 < n < n_cu:
        raise ops.DynError(f"DYN_CHAIN_CU_MASK={spec!r}: hole size must be in 1..{n_cu - 1}")
    hole = {(k + j * (n_cu // n)) % n_cu for j in range(n)} if strided else {(k * n + j) % n_cu for j in range(n)}
    words = (ctypes.c_uint32 * ((n_cu + 31) // 32))()
    for cu in range(n_cu):
        if cu not in hole:
            words[cu // 32] |= 1 << (cu % 32)
    out = ctypes.c_void_p()
    with torch.cuda.device(device):
        check(load().dyn_stream_create_cu_mask(words, len(words), ctypes.byref(out)), "dyn_stream_create_cu_mask")
    return torch.cuda.ExternalStream(out.value, device=device)


import os as _os

_CLEAN_COPY_FUSED_ATTN = _os.environ.get("DYN_CLEAN_FUSED_ATTN", "1") != "0"     # A/B switch
_CHAIN_STREAMS = {}
HOST_WAIT = [0.0]   # seconds the host spent blocked on the per-window pseudo-label ids (diagnostic)


def lockstep_group_sizes(n_items, R, n_chains):
    """How `n_items` recordings are cut into lockstep groups for `n_chains` group models of R replicas: as few groups as hold them, but — when
    there is more than one group — a multiple of the chain count, of sizes as equal as possible, so that the chains finish together.  20 recordings,
    R = 4, 2 chains: 4 4 3 3 3 3 (each chain 10) instead of 4 4 4 4 4, whose last group runs alone on the chip at the one-chain rate (measured:
    758 against 808 audio-s/s with two groups in flight)."""
    if n_items <= 0:
        return []
    G = -(-n_items // R)
    if n_chains > 1 and G > 1:
        G = min(n_items, -(-G // n_chains) * n_chains)
    base, extra = divmod(n_items, G)
    return [base + 1] * extra + [base] * (G - extra)


def dynamic_eval_many(args, models, specs, seq_len, overlap, tokenizer, **kw):
    """Several recordings in flight on ONE GPU from one host thread: each model replica in `models` owns a HIP stream and
    runs one recording at a time; the chains are advanced round-robin at their yield points, so the GEMMs of one chain fill
    the latency-bound stretches of the other (CTC scans, the pseudo-label round trip, short HBM-bound kernels).
    Recordings are independent (reference lib.py:494,636-637).  Returns the per-recording results in `specs` order."""
    device = models[0].device
    key = torch.device(device).index
    while len(_CHAIN_STREAMS.setdefault(key, [])) < len(models):    # streams are kept: the caching allocator's per-stream
        _CHAIN_STREAMS[key].append(_new_chain_stream(device, len(_CHAIN_STREAMS[key])))  # pools stay warm across calls (no hipMalloc in the loop)
    streams = _CHAIN_STREAMS[key][:len(models)]
    main = torch.cuda.current_stream(device)
    for st in streams:
        st.wait_stream(main)
    # Identical chains started together stay phase-locked (same kernels at the same time: when all of them are in their HBM-bound or
    # latency-bound stretches the matrix cores idle).  Chain k therefore starts `k * stagger` later, by a device-side delay on its stream.
    import os
    stagger_us = int(float(os.environ.get("DYN_CHAIN_STAGGER_MS", "0")) * 1000)
    if stagger_us > 0:
        from ._lib import check, load
        for k, st in enumerate(streams):
            if k:
                check(load().dyn_sleep_us(min(k * stagger_us, 2000000), st.cuda_stream), "dyn_sleep_us")
    pending = list(enumerate(specs))
    R = getattr(models[0], "R", 1)
    group_sizes = kw.pop('group_sizes', None)    # explicit sizes of the lockstep groups, in order (bench.py's prewarm); default: lockstep_group_sizes
    if R > 1:
        # lockstep groups: consecutive recordings of equal length share one group model (up to R at a time); what the group form does not
        # cover (lockstep_supported) cannot run on a group model at all, so it is refused here rather than silently run differently
        items, k = [], 0
        sizes = list(group_sizes) if group_sizes else lockstep_group_sizes(len(specs), R, len(models))
        while k < len(specs):
            n = n_plan = min(sizes.pop(0) if sizes else R, R, len(specs) - k)
            while n > 1 and not lockstep_supported(args, models[0], specs[k:k + n], kw.get('beam_search_fn'), kw.get('optimizer_state')):
                n -= 1          # e.g. several epochs: only recordings of equal length share a group
            if not lockstep_supported(args, models[0], specs[k:k + n], kw.get('beam_search_fn'), kw.get('optimizer_state')):
                raise ops.DynError("dynamic_eval_many: these arguments need the one-recording-per-model path (pass ungrouped models)")
            items.append((list(range(k, k + n)), specs[k:k + n]))
            k += n
            if n != n_plan:     # the plan no longer adds up: plan the rest again
                sizes = lockstep_group_sizes(len(specs) - k, R, len(models))
        pending = items
    results = [None] * len(specs)
    free, active = list(range(len(models)))[::-1], []
    try:
        _run_chains(args, models, streams, pending, results, free, active, seq_len, overlap, tokenizer, kw)
    finally:
        _reset_grad_samples(models)
    for st in streams:
        main.wait_stream(st)
    return results


def _run_chains(args, models, streams, pending, results, free, active, seq_len, overlap, tokenizer, kw):
    """Round-robin over the chains' generators (see dynamic_eval_many)."""
    while pending or active:
        while pending and free:
            ci = free.pop()
            idx, spec = pending.pop(0)
            if isinstance(idx, list):     # a lockstep group of recordings on a group model
                active.append([_dynamic_eval_group_gen(args, models[ci], spec, seq_len, overlap, tokenizer, **kw), ci, idx])
            else:
                active.append([_dynamic_eval_gen(args, models[ci], spec, seq_len, overlap, tokenizer, **kw), ci, idx])
        for item in list(active):
            gen, ci, idx = item
            with torch.cuda.stream(streams[ci]):
                try:
                    next(gen)
                except StopIteration as stop:
                    if isinstance(idx, list):
                        for j, res in zip(idx, stop.value):
                            results[j] = res
                    else:
                        results[idx] = stop.value
                    active.remove(item)
                    free.append(ci)


def _reset_grad_samples(models):
    for m in models:
        if _is_native(m):
            m.grad_samples = None


dynamic_eval = dynamic_eval_ctc_loss


def AWMC(*a, **kw):
    """Reference lcasr/lib.py:206-376 — implemented in awmc.py (same signature)."""
    from .awmc import AWMC as _impl
    return _impl(*a, **kw)


# ------------------------------------------------------------------------------------------------ shared CLI surface
def apply_args(parser, argv=None):
    """Reference lcasr/lib.py:1756-1787 (same flags, same free-form `-kwargs k=v` evaluated into args.__dict__)."""
    parser.add_argument('-c', '--checkpoint', type=str, default='', help='path to checkpoint')
    parser.add_argument('-split', '--split', type=str, default='test', help='test or dev split')
    parser.add_argument('-seq', '--seq_len', type=int, default=16384, help='-1 to use setting from config in checkpoint file')
    parser.add_argument('-o', '--overlap', type=int, default=14336, help='-1 to use setting from config in checkpoint file')
    parser.add_argument('-nv', '--not_verbose', action='store_true', help='verbose')
    parser.add_argument('-log', '--log', type=str, default='')
    parser.add_argument('-ds', '--dont_shuffle', action='store_true', help='dont shuffle')
    parser.add_argument('-epochs', '--epochs', type=int, default=1, help='epochs')
    parser.add_argument('-dfa', '--disable_flash_attention', action='store_true', help='disable flash attention')
    parser.add_argument('-beamsearch', '--beamsearch', action='store_true', help='use beam search')
    parser.add_argument('-kwargs', '--kwargs', nargs='+', help='kwargs')
    parser.add_argument('-awmc', '--awmc', action='store_true', help='Use AWMC instead of dynamic eval')
    parser.add_argument('--consistency', '--consistency', action='store_true', help='Use consistency training')
    parser.add_argument('--freeze_subsampling', action='store_true')
    parser.add_argument('--freeze_all_but_last_block_and_head', action='store_true')
    parser.add_argument('--train_subsampling_only', action='store_true')
    args = parser.parse_args(argv)
    if args.kwargs is None:
        args.kwargs = []
    for kwarg in args.kwargs:
        key, value = kwarg.split('=')
        args.__dict__[key] = eval(value)  # same free-form override as the reference (lib.py:1780)
        print(f'Overriding {key} to {value}')
    args.shuffle = not args.dont_shuffle
    args.verbose = not args.not_verbose
    return args
