"""ORACLE (test infrastructure only).  CPU restatement of the reference's optional Loop-A augmentations, following
reference lcasr/lib.py line by line: frame_shuffle (:81-84), add_random_noise (:379-382), cutout (:384-417).
These functions ARE in the reference (pure torch, no un-vendored dependency), so this restatement is pinned by the
reference source itself; it differs only in taking/returning a [1, F, T] CPU tensor explicitly."""
import torch


def frame_shuffle(spec, time_dimension=False, freq_dimension=False):
    if time_dimension:
        spec = spec[:, :, torch.randperm(spec.shape[-1])]
    if freq_dimension:
        spec = spec[:, torch.randperm(spec.shape[-2]), :]
    return spec


def add_random_noise(spec, noise_factor):
    if noise_factor == 0:
        return spec
    noise = torch.normal(0, std=spec.std(), size=spec.shape)
    return spec + noise * noise_factor


def cutout(spec, seq_len, cutout_val='mean', num_rectangles=5, max_width=100, max_height=10):
    if num_rectangles == 0:
        return spec
    spec_n = spec.shape[-1]
    num_rectangles = int(num_rectangles * (spec_n / seq_len))
    widths = torch.randint(1, max_width, (num_rectangles,))
    heights = torch.randint(1, max_height, (num_rectangles,))
    sx = torch.randint(0, spec.shape[-1], (num_rectangles,))
    ex = (sx + widths).clamp(max=spec.shape[-1])
    sy = torch.randint(0, spec.shape[-2], (num_rectangles,))
    ey = (sy + heights).clamp(max=spec.shape[-2])
    if cutout_val == 'mean_recording':
        mask_value = spec.mean()
    elif cutout_val == 'mean':
        mask_values = [spec[:, sy[i]:ey[i], sx[i]:ex[i]].mean() for i in range(num_rectangles)]
    for i in range(num_rectangles):
        if cutout_val == 'mean':
            spec[:, sy[i]:ey[i], sx[i]:ex[i]] = mask_values[i]
        elif cutout_val == 'mean_recording':
            spec[:, sy[i]:ey[i], sx[i]:ex[i]] = mask_value
        elif cutout_val == 'zero':
            spec[:, sy[i]:ey[i], sx[i]:ex[i]].zero_()
    return spec
