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
