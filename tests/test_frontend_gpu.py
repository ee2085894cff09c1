"""Log-mel front end: HIP (STFT as an fp32-MFMA GEMM over overlapping rows) vs the torch.stft oracle (float64)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [16000, 16000 * 7 + 123, 800])
def test_logmel_matches_oracle(cuda, n):
    from dynamic_asr_eval_amd.frontend import LogMel
    from oracle.logmel_ref import logmel_ref
    g = torch.Generator().manual_seed(n)
    t = torch.arange(n) / 16000.0
    wav = 0.1 * torch.randn(n, generator=g) + 0.3 * torch.sin(2 * torch.pi * 440.0 * t) + 0.2 * torch.sin(2 * torch.pi * 3100.0 * t)
    for normalize in (False, True):
        ref = logmel_ref(wav, normalize=normalize)
        out = LogMel(cuda, normalize=normalize)(wav)
        assert out.shape == ref.shape == (1, 80, 1 + n // 160)
        diff = (out.cpu().double() - ref).abs()
        if normalize:   # the normalisation divides by the per-bin std (tiny for the stationary tone bins): compare in log-mel units
            diff = diff * logmel_ref(wav, normalize=False).std(-1, keepdim=True)
        err = diff.max().item()
        assert err < 1e-3, (normalize, err)      # fp32 GEMM over K = 400 vs float64 FFT, after a log


def test_tedlium_segment_zeroing(cuda):
    """reference lcasr/tedlium/run.py:91-96: `ignore_time_segment_in_scoring` spans are blanked; > 32 segments (chunked kernel
    arguments), a segment running past the end, an empty one."""
    from dynamic_asr_eval_amd.frontend import zero_out_spectogram
    from oracle.logmel_ref import zero_out_spectogram_ref
    g = torch.Generator().manual_seed(5)
    spec = torch.randn(1, 80, 6000, generator=g)
    segs = [{'start': 0.37 * k + 0.05, 'end': 0.37 * k + 0.05 + 0.11 * (k % 3)} for k in range(40)] + [{'start': 59.5, 'end': 75.0}]
    ref = zero_out_spectogram_ref(spec, segs)
    out = zero_out_spectogram(spec.to(cuda).contiguous(), segs)
    assert torch.equal(out.cpu(), ref)
    assert (ref == 0).any() and not (ref == 0).all()


def test_chime6_channel_average(cuda):
    """reference lcasr/chime6/run.py:46-70: channels of unequal length -> log-mel each -> trim -> mean -> renormalise."""
    from dynamic_asr_eval_amd.frontend import combine_channels
    from oracle.logmel_ref import combine_channels_ref, logmel_ref
    g = torch.Generator().manual_seed(8)
    lens = [16000 * 6, 16000 * 6 - 517, 16000 * 5 + 999, 16000 * 6]
    t = torch.arange(max(lens)) / 16000.0
    base = 0.3 * torch.sin(2 * torch.pi * 310.0 * t) * (1 + 0.5 * torch.sin(2 * torch.pi * 1.5 * t))
    wavs = [(base + 0.05 * torch.randn(max(lens), generator=g))[:n] for n in lens]
    ref = combine_channels_ref(wavs, 0.5, 5.25)
    out = combine_channels(wavs, 0.5, 5.25, device=cuda)
    assert out.shape == ref.shape == (1, 80, 475)
    scale = torch.stack([logmel_ref(torch.nn.functional.pad(w, (0, max(lens) - w.numel())), normalize=False)[:, :, 50:525] for w in wavs]).mean(0).std(-1, keepdim=True)
    err = ((out.cpu().double() - ref).abs() * scale).max().item()      # in log-mel units (see test_logmel_matches_oracle)
    assert err < 1e-3, err
    m = out.mean(-1).abs().max().item(); sd = (out.std(-1) - 1).abs().max().item()
    assert m < 1e-4 and sd < 1e-3
