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
        """window: contiguous CUDA [F, T]; masks = ((f_starts, f_widths), (t_starts, t_widths))."""
        (f0, fw), (t0, tw) = masks
        dev = window.device
        if f0:
            ops.specaug_freqmask(window, torch.tensor(f0, dtype=torch.int32, device=dev),
                                 torch.tensor(fw, dtype=torch.int32, device=dev), fill_value)
        if t0:
            ops.specaug_timemask(window, torch.tensor(t0, dtype=torch.int32, device=dev),
                                 torch.tensor(tw, dtype=torch.int32, device=dev), fill_value)
        return window
