"""SpecAugment for the augmented copy of each window (reference lcasr/lib.py:102-112,499,541).

The reference takes `SpecAugment` from the un-vendored `lcasr.utils.augmentation`; only its configuration surface is
visible in the reference (`n_time_masks, n_freq_masks, freq_mask_param, time_mask_param, min_p, zero_masking`,
lib.py:104-111; canonical run: 6 frequency masks of at most 34 bins, no time masks —
earnings_finetune/lcasr160rb1.yaml:73-81).  Mask rule DEFINED here (parity unpinned against upstream):
  width ~ U{0..param}, start ~ U{0..size-width}, drawn with the caller's torch CPU RNG in mask order (frequency
  masks first, then time masks); masked cells are set to 0 if `zero_masking` else to the mean of the window.
  `time_mask_param == -1` means proportional masks: width ~ U{0..max(1, int(min_p * T))}.
The masking itself is a HIP kernel on the device-resident window; only the (start, width) pairs come from the host."""
import torch

from . import ops


def draw_masks(n_masks, param, size, generator=None):
    """Host-side draw of (starts, widths), torch RNG — shared rule for the product and the oracle."""
    starts, widths = [], []
    for _ in range(int(n_masks)):
        w = int(torch.randint(0, int(param) + 1, (1,), generator=generator).item()) if param > 0 else 0
        w = min(w, size)
        s = int(torch.randint(0, size - w + 1, (1,), generator=generator).item())
        starts.append(s)
        widths.append(w)
    return starts, widths


class SpecAugment:
    def __init__(self, n_time_masks=0, n_freq_masks=0, freq_mask_param=42, time_mask_param=-1, min_p=0.05,
                 zero_masking=False, **_):
        self.n_time_masks, self.n_freq_masks = int(n_time_masks), int(n_freq_masks)
        self.freq_mask_param, self.time_mask_param = int(freq_mask_param), int(time_mask_param)
        self.min_p, self.zero_masking = float(min_p), bool(zero_masking)

    def draw(self, F, T, generator=None):
        fm = draw_masks(self.n_freq_masks, self.freq_mask_param, F, generator)
        tparam = self.time_mask_param if self.time_mask_param >= 0 else max(1, int(self.min_p * T))
        tm = draw_masks(self.n_time_masks, tparam, T, generator)
        return fm, tm

    def apply(self, window, masks, fill_value):
        """window: contiguous CUDA [F, T]; masks = ((f_starts, f_widths), (t_starts, t_widths)); fill_value: float or a
        1-element device tensor.  Mask positions travel as kernel arguments (no H2D copy, no host stall)."""
        import ctypes
        from ._lib import check, load
        F, T = window.shape
        vdev = fill_value.data_ptr() if isinstance(fill_value, torch.Tensor) else 0
        val = 0.0 if vdev else float(fill_value)
        for along_time, (st, wd) in ((0, masks[0]), (1, masks[1])):
            for i in range(0, len(st), 32):
                a = (ctypes.c_int32 * len(st[i:i + 32]))(*st[i:i + 32])
                b = (ctypes.c_int32 * len(wd[i:i + 32]))(*wd[i:i + 32])
                check(load().dyn_specaug_mask_args(window.data_ptr(), F, T, ctypes.addressof(a), ctypes.addressof(b), len(a), along_time,
                                                   val, vdev, torch.cuda.current_stream().cuda_stream), "dyn_specaug_mask_args")
        return window


# ---------------------------------------------------------------------------------------------------------------------
# Optional augmentations of Loop A (reference lcasr/lib.py:542-544).  Random draws use the caller's torch CPU RNG in the
# reference's order; the data movement / arithmetic runs on the device window ([F, T], contiguous, modified in place).
def _moments(window):
    from ._lib import check, load
    out = torch.empty(3, device=window.device, dtype=torch.float32)
    ws = ops.workspace(window.device)
    check(load().dyn_moments(window.data_ptr(), window.numel(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                             torch.cuda.current_stream().cuda_stream), "dyn_moments")
    s, mean, std = out.cpu().tolist()
    return s, mean, std


def frame_shuffle(window, time_dimension=False, freq_dimension=False):
    """reference lib.py:81-84"""
    from ._lib import check, load
    F, T = window.shape
    for along_time, n, on in ((1, T, time_dimension), (0, F, freq_dimension)):
        if not on:
            continue
        perm = torch.randperm(n).to(torch.int32).to(window.device)
        src = window.clone()
        check(load().dyn_gather_frames(src.data_ptr(), perm.data_ptr(), window.data_ptr(), F, T, along_time,
                                       torch.cuda.current_stream().cuda_stream), "dyn_gather_frames")
    return window


def add_random_noise(window, noise_factor):
    """reference lib.py:379-382: spec + N(0, spec.std()) * noise_factor (noise drawn on the host like the reference)."""
    if noise_factor == 0:
        return window
    _, _, std = _moments(window)
    noise = torch.normal(0, std=std, size=tuple(window.shape)).to(window.device)
    ops.axpby(noise, window, a=float(noise_factor), b=1.0)
    return window


def cutout(window, seq_len, cutout_val='mean', num_rectangles=5, max_width=100, max_height=10):
    """reference lib.py:384-417 (same draw order: widths, heights, start x, start y)."""
    from ._lib import check, load
    if num_rectangles == 0:
        return window
    F, T = window.shape
    num_rectangles = int(num_rectangles * (T / seq_len))
    if num_rectangles <= 0:
        return window
    widths = torch.randint(1, max_width, (num_rectangles,))
    heights = torch.randint(1, max_height, (num_rectangles,))
    sx = torch.randint(0, T, (num_rectangles,))
    ex = (sx + widths).clamp(max=T)
    sy = torch.randint(0, F, (num_rectangles,))
    ey = (sy + heights).clamp(max=F)
    rects = torch.stack([sy, ey, sx, ex], 1).to(torch.int32).contiguous().to(window.device)
    means = torch.empty(num_rectangles, device=window.device, dtype=torch.float32)
    if cutout_val == 'mean':
        mode, value = 1, 0.0
    elif cutout_val == 'mean_recording':
        mode, value = 2, _moments(window)[1]
    elif cutout_val == 'zero':
        mode, value = 0, 0.0
    else:
        raise ValueError(f"unknown cutout value {cutout_val!r}")
    check(load().dyn_cutout(window.data_ptr(), F, T, rects.data_ptr(), num_rectangles, mode, value, means.data_ptr(),
                            torch.cuda.current_stream().cuda_stream), "dyn_cutout")
    return window
