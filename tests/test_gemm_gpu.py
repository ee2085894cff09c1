"""GEMM parity: HIP fp32-MFMA kernel (through the C-ABI) vs a float64 CPU product of the same operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, b, ta, tb):
    A = a.double().cpu(); B = b.double().cpu()
    if ta: A = A.transpose(-1, -2)
    if tb: B = B.transpose(-1, -2)
    return A @ B


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 96), (100, 70, 50), (33, 129, 7), (1, 5, 3),
                                   (512, 768, 768), (4096, 768, 3072)])
def test_gemm_layouts(cuda, ta, tb, M, N, K):
    from dynamic_asr_eval_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(cuda)
    b = torch.randn((N, K) if tb else (K, N), generator=g).to(cuda)
    c = torch.full((M, N), float("nan"), device=cuda)
    ops.gemm(a, b, c, trans_a=ta, trans_b=tb, M=M, N=N, K=K, lda=a.shape[1], ldb=b.shape[1], ldc=N)
    ref = _ref(a, b, ta, tb)
    err = (c.double().cpu() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, K ** 0.5) * 4, err


def test_gemm_identity_asymmetric(cuda):
    """A = I with an asymmetric B catches a transposed C write (cdna_hip_programming.md §3)."""
    from dynamic_asr_eval_amd import ops
    n = 128
    a = torch.eye(n, device=cuda)
    b = (torch.arange(n * n, device=cuda, dtype=torch.float32).reshape(n, n) % 251) - 100.0
    c = torch.empty(n, n, device=cuda)
    ops.gemm(a, b, c, M=n, N=n, K=n, lda=n, ldb=n, ldc=n)
    assert torch.equal(c, b)


def test_gemm_alpha_beta_bias_splitk(cuda):
    from dynamic_asr_eval_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 96, 160, 4096
    a = torch.randn(K, M, generator=g).to(cuda)       # dW = dY^T X shape family
    b = torch.randn(K, N, generator=g).to(cuda)
    c0 = torch.randn(M, N, generator=g).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    for split in (0, 1, 4, 7):
        c = c0.clone()
        ops.gemm(a, b, c, trans_a=True, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, alpha=0.5, beta=2.0, bias=bias,
                 split_k=split)
        ref = 0.5 * (a.double().cpu().T @ b.double().cpu()) + 2.0 * c0.double().cpu() + bias.double().cpu()
        assert (c.double().cpu() - ref).abs().max().item() < 2e-3, split


def test_gemm_two_level_batch(cuda):
    """Attention-style addressing: heads inside a [B, T, H*D] activation, scores [B, H, T, T]."""
    from dynamic_asr_eval_amd import ops
    g = torch.Generator().manual_seed(9)
    B, T, H, D = 2, 160, 3, 64
    q = torch.randn(B, T, H * D, generator=g).to(cuda)
    k = torch.randn(B, T, H * D, generator=g).to(cuda)
    s = torch.empty(B, H, T, T, device=cuda)
    ops.gemm(q, k, s, trans_b=True, M=T, N=T, K=D, lda=H * D, ldb=H * D, ldc=T, nb1=B, nb2=H,
             sa=(T * H * D, D), sb=(T * H * D, D), sc=(H * T * T, T * T), alpha=0.125)
    qh = q.view(B, T, H, D).permute(0, 2, 1, 3).double().cpu()
    kh = k.view(B, T, H, D).permute(0, 2, 1, 3).double().cpu()
    ref = 0.125 * qh @ kh.transpose(-1, -2)
    assert (s.double().cpu() - ref).abs().max().item() < 1e-4
