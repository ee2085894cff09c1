"""AWMC test-time adaptation (anchor / leader EMA teachers + noisy student) with the reference's signature
`AWMC(args, model, spec, seq_len, overlap, tokenizer, use_tqdm, optim, optimizer_state, beam_search_fn, return_params)`
(reference lcasr/lib.py:206-376; selected by `-awmc`, run_dynamic_eval_full.py:67-68), on the same HIP kernels as
dynamic_eval.  Per window and epoch (lib.py:280-346): anchor forward (EMA decay 1.0 = the original weights) and leader
forward (EMA decay `ema_decay`, default 0.999) give two greedy pseudo-label banks; the augmented window goes through the
student with grad; CTC loss against BOTH banks / (N*B*2); one optimiser step; leader EMA update; after the last epoch a
clean no-grad forward feeds the on-device stitch.

`torch_ema.ExponentialMovingAverage` (un-vendored, absent) is restated: shadow -= (1 - d) * (shadow - param) with the
warm-up d = min(decay, (1 + n) / (10 + n)) of torch_ema's default `use_num_updates=True`; parity unpinned against it.
EMA shadows are flat HBM buffers; swapping them in and out of the model is three device-to-device copies."""
import time

import torch

from . import ops
from .augment import SpecAugment
from .decoding import GreedyCTCDecoder
from .optim import MADGRAD

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, **_):
        return x


class FlatEMA:
    """torch_ema.ExponentialMovingAverage over the model's flat parameter buffer."""

    def __init__(self, flat_params, decay):
        self.decay, self.num_updates = float(decay), 0
        self.shadow = flat_params.clone()
        self._saved = None

    def update(self, flat_params):
        self.num_updates += 1
        d = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        ops.axpby(flat_params, self.shadow, a=1.0 - d, b=d)   # shadow = d * shadow + (1 - d) * param

    def swap_in(self, flat_params):
        self._saved = flat_params.clone()
        flat_params.copy_(self.shadow)

    def swap_out(self, flat_params):
        flat_params.copy_(self._saved)
        self._saved = None


def AWMC(args, model, spec, seq_len, overlap, tokenizer, use_tqdm=True, optim=MADGRAD, optimizer_state=None,
         beam_search_fn=None, return_params=False, return_device=False):
    from . import lib
    assert beam_search_fn is None, 'Beam search function not implemented for AWMC'
    if not lib._is_native(model):
        raise ops.DynError("AWMC: only the HIP SCConformerXL is supported (flat EMA buffers)")
    spec_augment_config = lib.get_specaugment_config_from_args(args)
    lr_args = lib.get_lr_args_from_args(args)
    fs = lib.get_frame_shuffle_config_from_args(args)
    if fs['time_dimension'] or fs['freq_dimension']:
        lib._unsupported("frame_shuffle")
    device = model.device
    spec_n = spec.shape[-1]
    downsampling_factor = args.config['model']['subsampling_factor']
    seq_len = seq_len if seq_len != -1 else args.config['audio_chunking']['size']
    original_flat = model.flat_params.clone()
    frozen_before = set(model.frozen)
    if args.__dict__.get('bitfit', False):                          # reference lib.py:234-235
        lib.bitfit(model)
    if args.__dict__.get('freeze_subsampling', False):
        lib.freeze_subsampling(model)
    if args.__dict__.get('freeze_all_but_last_block_and_head', False):
        lib.freeze_all_but_last_block_and_head(model)
    if args.__dict__.get('train_subsampling_only', False):
        lib.train_subsampling_only(model)

    model.train()
    flat = model.flat_params
    ema_leader = FlatEMA(flat, args.__dict__.get('ema_decay', 0.999)); ema_leader.update(flat)
    ema_anchor = FlatEMA(flat, 1.0); ema_anchor.update(flat)
    num_classes = model.decoder.num_classes
    blank = num_classes - 1
    optimizer = optim(model.parameters(), **lr_args)
    if optimizer_state is not None:
        optimizer.load_state_dict(optimizer_state)
    decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=blank, device=device)
    augmentation = SpecAugment(**spec_augment_config)
    fixed_masks = args.__dict__.get('spec_augment_fixed_masks', None)
    if seq_len > spec_n:
        seq_len, overlap = spec_n, 0
    else:
        overlap = overlap if overlap != -1 else args.config['audio_chunking']['overlap']
    assert args.config['training'].get("max_seq_len", 0) == 0, 'caching is not used anymore'
    assert overlap / downsampling_factor == overlap // downsampling_factor, 'Overlap must be a multiple of the downsampling factor'
    epochs = args.__dict__.get('epochs', 1)
    print_runtimes = args.__dict__.get('print_runtimes', False)
    spec_dev = spec.to(device=device, dtype=torch.float32)
    Fq = spec_dev.shape[1]
    acc = torch.zeros(spec_n // 4 + seq_len, num_classes, device=device, dtype=torch.float32)
    cnt = torch.zeros(spec_n // 4 + seq_len, device=device, dtype=torch.float32)
    pos = end = 0
    training_data, training_keys = lib.prepare_chunks(spec_dev, seq_len, overlap)
    training_keys = list(training_data.keys())
    pbar = tqdm(training_keys) if use_tqdm else training_keys
    stime = time.time()

    def greedy_targets(window):
        with torch.no_grad():
            out = model(audio_signal=window)
        return tokenizer.encode(decoder(out['final_posteriors'][-1]))

    model.eval()
    for i in pbar:
        label_bank = [None, None]
        clean = training_data[i].contiguous()        # [1, F, u_len]
        u_len = clean.shape[-1]
        for j in range(epochs):
            if j == 0:
                ema_anchor.swap_in(flat); label_bank[0] = greedy_targets(clean); ema_anchor.swap_out(flat)
            ema_leader.swap_in(flat); label_bank[1] = greedy_targets(clean); ema_leader.swap_out(flat)
            noisy = clean.clone()
            masks = fixed_masks[i] if fixed_masks is not None else augmentation.draw(Fq, u_len)
            if masks[0][0] or masks[1][0]:
                augmentation.apply(noisy[0], masks, lib._window_fill_value(noisy[0], augmentation.zero_masking))
            with torch.enable_grad():
                out = model(audio_signal=noisy)
            post = out['final_posteriors']           # [1, N, C]
            labels = [el for el in label_bank if len(el) > 0] or [[]]
            nb = len(labels)
            S_max = max(1, max(len(el) for el in labels))
            tgt = torch.zeros(nb, S_max, dtype=torch.int32)
            for b, el in enumerate(labels):
                tgt[b, :len(el)] = torch.tensor(el, dtype=torch.int32)
            N, B = post.shape[1], post.shape[0]
            total_tokens_in_loss = N * B * 2
            lp_rep = post.expand(nb, N, num_classes).contiguous()   # posteriors repeated per label bank (lib.py:325)
            ilen = torch.full((nb,), N, dtype=torch.int32, device=device)
            tlen = torch.tensor([len(el) for el in labels], dtype=torch.int32, device=device)
            _, _, g = ops.ctc_loss(lp_rep, tgt.to(device), ilen, tlen, blank, reduction="sum", grad_scale=1.0 / total_tokens_in_loss)
            grad = g[0:1].contiguous()
            for b in range(1, nb):                    # backward of .repeat(): the banks' gradients add up
                ops.axpby(g[b], grad[0], a=1.0, b=1.0)
            optimizer.zero_grad()
            model.backward(grad)
            optimizer.step()
            ema_leader.update(flat)
            if j == epochs - 1:
                with torch.no_grad():
                    lp = model(audio_signal=clean)['final_posteriors'][0]
                ds_len = lp.shape[0]
                overlap_ds = int(overlap / (u_len / ds_len))
                pos -= overlap_ds if i != 0 else 0
                ops.stitch_accumulate(lp, acc, cnt, pos)
                pos += ds_len
                end = max(end, pos)
    if print_runtimes:
        torch.cuda.synchronize(device)
        print(f'Runtime: {time.time() - stime}')
    logits_dev = ops.stitch_finalize(acc, cnt, end)
    if return_params:
        updated = [p.clone().detach().cpu() for p in model.parameters()]
    model.flat_params.copy_(original_flat)
    model.frozen = frozen_before
    logits = logits_dev if return_device else logits_dev.cpu().numpy()
    return logits if not return_params else (logits, updated)
