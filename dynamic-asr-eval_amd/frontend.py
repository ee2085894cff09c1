"""Log-mel front end on the GPU: 16 kHz mono waveform -> spec [1, 80, T] (100 frames / s), the tensor every
`process_fn` of the reference hands to eval_fn (reference lcasr/run_dynamic_eval_full.py:84; adapters call the
un-vendored `lcasr.utils.audio_tools.processing_chain`, e.g. lcasr/earnings22/run.py:61).

Constants DEFINED here (the upstream implementation is absent; parity unpinned, the oracle oracle/logmel_ref.py restates
the same definition with torch.stft): n_fft 512, Hann(400, periodic) centred in the frame, hop 160, center=True with
reflect padding, power spectrum, 80 HTK-mel triangular filters over 0-8000 Hz (torchaudio `melscale_fbanks` rule,
norm=None), log(mel + 1e-6), then per-bin (mean, unbiased std) normalisation over the recording.
The STFT is one fp32-MFMA GEMM over overlapping rows of the padded signal (lda = hop); see csrc/logmel.hip."""
import math

import torch

from . import ops
from ._lib import check, load

SAMPLE_RATE, N_FFT, WIN, HOP, N_MELS, KP = 16000, 512, 400, 160, 80, 260   # KP = 257 bins padded to a multiple of 4


def mel_filterbank(n_freqs=N_FFT // 2 + 1, f_min=0.0, f_max=8000.0, n_mels=N_MELS, sample_rate=SAMPLE_RATE):
    """[n_freqs, n_mels] triangular HTK-mel filters (torchaudio.functional.melscale_fbanks, norm=None), float64."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


class LogMel:
    def __init__(self, device="cuda:0", eps=1e-6, normalize=True):
        self.device, self.eps, self.normalize = torch.device(device), float(eps), bool(normalize)
        n = torch.arange(WIN, dtype=torch.float64)
        w = 0.5 - 0.5 * torch.cos(2 * math.pi * n / WIN)                      # torch.hann_window(400, periodic=True)
        k = torch.arange(N_FFT // 2 + 1, dtype=torch.float64)
        ang = 2 * math.pi * k[None, :] * (n[:, None] + (N_FFT - WIN) // 2) / N_FFT
        basis = torch.zeros(WIN, 2 * KP, dtype=torch.float64)
        basis[:, :N_FFT // 2 + 1] = w[:, None] * torch.cos(ang)
        basis[:, KP:KP + N_FFT // 2 + 1] = -w[:, None] * torch.sin(ang)
        self.basis = basis.float().to(self.device).contiguous()                # [400, 520]
        fb = torch.zeros(KP, N_MELS, dtype=torch.float64)
        fb[:N_FFT // 2 + 1] = mel_filterbank()
        self.fb = fb.float().to(self.device).contiguous()                      # [260, 80]

    def __call__(self, waveform):
        """waveform: 1-D float tensor (host or device) -> spec [1, 80, T] on the device, T = 1 + len // 160."""
        x = torch.as_tensor(waveform, dtype=torch.float32).reshape(-1).to(self.device).contiguous()
        n = x.numel()
        pad = N_FFT // 2
        T = 1 + n // HOP
        st = torch.cuda.current_stream().cuda_stream
        xpad = torch.empty(n + 2 * pad, device=self.device, dtype=torch.float32)
        check(load().dyn_reflect_pad(x.data_ptr(), xpad.data_ptr(), n, pad, st), "dyn_reflect_pad")
        reim = torch.empty(T, 2 * KP, device=self.device, dtype=torch.float32)
        # frame t = xpad[t*160 + 56 : t*160 + 456]  (the 400-tap window sits at offset 56 of the 512-point frame)
        ops.gemm(xpad, self.basis, reim, M=T, N=2 * KP, K=WIN, lda=HOP, ldb=2 * KP, ldc=2 * KP, a_off=(N_FFT - WIN) // 2)
        power = torch.empty(T, KP, device=self.device, dtype=torch.float32)
        check(load().dyn_stft_power(reim.data_ptr(), power.data_ptr(), T, KP, st), "dyn_stft_power")
        mel = torch.empty(T, N_MELS, device=self.device, dtype=torch.float32)
        ops.gemm(power, self.fb, mel, M=T, N=N_MELS, K=KP, lda=KP, ldb=N_MELS, ldc=N_MELS)
        out = torch.empty(1, N_MELS, T, device=self.device, dtype=torch.float32)
        ws = ops.workspace(self.device)
        check(load().dyn_logmel_finish(mel.data_ptr(), out.data_ptr(), T, N_MELS, self.eps, int(self.normalize), ws.data_ptr(),
                                       ws.numel(), st), "dyn_logmel_finish")
        return out


def processing_chain(waveform, device="cuda:0"):
    """Same role as upstream `lcasr.utils.audio_tools.processing_chain`: waveform -> normalised log-mel [1, 80, T]."""
    return LogMel(device)(waveform)


def total_frames(seconds):
    """Seconds -> spectrogram frames (upstream `audio_tools.total_frames`; 100 frames / s, DEFINED here as int(s * 100))."""
    return int(float(seconds) * 100)


def zero_out_spectogram(spec, remove_timings):
    """TEDLIUM `ignore_time_segment_in_scoring` segments are blanked before eval (reference lcasr/tedlium/run.py:91-96 calls
    the un-vendored `lcasr.eval.utils.zero_out_spectogram`; rule DEFINED here: spec[:, :, frames(start):frames(end)] = 0).
    spec: CUDA [1, F, T], modified in place by the time-mask kernel (segments travel as kernel arguments)."""
    import ctypes
    F, T = spec.shape[-2], spec.shape[-1]
    if spec.dim() != 3 or spec.shape[0] != 1 or not spec.is_cuda or not spec.is_contiguous():
        raise ops.DynError("zero_out_spectogram expects a contiguous CUDA [1, F, T] spectrogram")
    st, wd = [], []
    for seg in remove_timings:
        a, b = min(total_frames(seg['start']), T), min(total_frames(seg['end']), T)
        if b > a:
            st.append(a); wd.append(b - a)
    for i in range(0, len(st), 32):
        a = (ctypes.c_int32 * len(st[i:i + 32]))(*st[i:i + 32])
        b = (ctypes.c_int32 * len(wd[i:i + 32]))(*wd[i:i + 32])
        check(load().dyn_specaug_mask_args(spec.data_ptr(), F, T, ctypes.addressof(a), ctypes.addressof(b), len(a), 1, 0.0, 0,
                                           torch.cuda.current_stream().cuda_stream), "dyn_specaug_mask_args")
    return spec


def combine_channels(waveforms, stime, etime, device="cuda:0"):
    """CHiME-6 array combination (reference lcasr/chime6/run.py:46-70): every channel -> un-normalised log-mel, right-padded
    to the longest channel, trimmed to [frames(stime), frames(etime)), averaged over channels, renormalised per bin
    ((x - mean) / unbiased std over time).  -> CUDA [1, 80, T']"""
    fe = LogMel(device, normalize=False)
    n = max(int(torch.as_tensor(w).numel()) for w in waveforms)
    a, b = total_frames(stime), total_frames(etime)
    acc = None
    for w in waveforms:
        x = torch.as_tensor(w, dtype=torch.float32).reshape(-1)
        if x.numel() < n:
            x = torch.nn.functional.pad(x, (0, n - x.numel()))
        spec = fe(x)[:, :, a:b].contiguous()
        if acc is None:
            acc = torch.zeros_like(spec)
        ops.axpby(spec, acc, a=1.0 / len(waveforms), b=1.0)
    check(load().dyn_rownorm(acc.data_ptr(), acc.data_ptr(), acc.shape[1], acc.shape[2], torch.cuda.current_stream().cuda_stream),
          "dyn_rownorm")
    return acc
