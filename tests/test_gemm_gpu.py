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


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(100, 72, 64), (67, 129, 32), (1992, 200, 128), (260, 4, 96), (4, 260, 32)])
@pytest.mark.parametrize("tile", [(128, 128, 1), (64, 64, 1), (128, 64, 3), (64, 128, 1), (256, 128, 1), (256, 128, 2)])
def test_gemm_direct_to_lds_edges(cuda, ta, tb, M, N, K, tile):
    """K % 32 == 0 and 16-B aligned operands select the direct-to-LDS kernel (when the row-contiguous operands have a row
    count that is a multiple of 4; otherwise the register-staged kernel): ragged M/N edges are clamped loads there, so check
    every tile shape against fp64 AND that nothing outside C[:M, :N] is touched (ldc > N, sentinel rows after M)."""
    from dynamic_asr_eval_amd import ops
    g = torch.Generator().manual_seed(M * 5 + N * 11 + K + tile[0])
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(cuda)
    b = torch.randn((N, K) if tb else (K, N), generator=g).to(cuda)
    ldc = N + 4
    c = torch.full((M + 3, ldc), 7.25, device=cuda)
    ops.gemm(a, b, c, trans_a=ta, trans_b=tb, M=M, N=N, K=K, lda=a.shape[1], ldb=b.shape[1], ldc=ldc, force=tile)
    ref = _ref(a, b, ta, tb)
    got = c.cpu()
    err = (got[:M, :N].double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, K ** 0.5) * 4, err
    assert torch.all(got[M:] == 7.25) and torch.all(got[:, N:] == 7.25), "wrote outside the M x N block"


def test_gemm_direct_to_lds_matches_register_staged_bitwise(cuda):
    """Both staging paths feed the MFMAs the same operands in the same k order: the results are bit-identical.  K = 96 runs
    the direct-to-LDS kernel; the same product as two K = 48 halves accumulated (beta = 1) cannot be compared bitwise, so the
    register-staged path is forced instead by a 4-B misaligned view of the same data (alignment is part of the dispatch)."""
    from dynamic_asr_eval_amd import ops
    g = torch.Generator().manual_seed(9)
    M, N, K = 300, 200, 96
    a = torch.randn(M, K, generator=g).to(cuda)
    b = torch.randn(N, K, generator=g).to(cuda)
    c1 = torch.empty(M, N, device=cuda); c2 = torch.empty(M, N, device=cuda)
    ops.gemm(a, b, c1, trans_b=True, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, force=(128, 128, 1))
    buf_a = torch.empty(M * K + 1, device=cuda); buf_a[1:].copy_(a.flatten())
    buf_b = torch.empty(N * K + 1, device=cuda); buf_b[1:].copy_(b.flatten())
    ops.gemm(buf_a, buf_b, c2, trans_b=True, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, a_off=1, b_off=1, force=(128, 128, 1))
    assert torch.equal(c1, c2)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_256x128_tile_is_bit_identical_to_128x128(cuda, ta, tb):
    """r04: the 8-wave 256x128 tile (direct-to-LDS staging only; 96 KB of LDS) gives every output element the same fmaf chain over k as the
    4-wave tiles: bit-identical results, batched over two levels with a bias, a residual and alpha / beta, ragged edges, tail slicing."""
    from dynamic_asr_eval_amd import ops
    g = torch.Generator().manual_seed(77 + 2 * ta + tb)
    for (M, N, K, nb1, nb2) in ((2048, 768, 768, 2, 3), (1992, 3072, 768, 1, 2), (700, 260, 96, 1, 1)):
        batch = nb1 * nb2
        a = torch.randn((batch, K, M) if ta else (batch, M, K), generator=g).to(cuda)
        b = torch.randn((nb2, N, K) if tb else (nb2, K, N), generator=g).to(cuda)      # the second batch level steps through the weights
        bias = torch.randn(nb2, N, generator=g).to(cuda)
        res = torch.randn(batch, M, N, generator=g).to(cuda)
        outs = []
        for tile in ((128, 128, 1), (256, 128, 1), (256, 128, 3)):
            c = torch.empty(batch, M, N, device=cuda)
            ops.gemm(a, b, c, trans_a=ta, trans_b=tb, M=M, N=N, K=K, lda=a.shape[2], ldb=b.shape[2], ldc=N, nb1=nb1, nb2=nb2,
                     sa=(nb2 * a.shape[1] * a.shape[2], a.shape[1] * a.shape[2]), sb=(0, b.shape[1] * b.shape[2]), sc=(nb2 * M * N, M * N),
                     bias=bias, sbias=(0, N), alpha=0.5, beta=1.0, c_in=res, force=tile)
            outs.append(c)
        assert torch.equal(outs[0], outs[1]), (M, N, K, "256x128 differs from 128x128")
        assert torch.equal(outs[0], outs[2]) or (outs[0] - outs[2]).abs().max().item() < 1e-4 * K ** 0.5, (M, N, K, "tail-sliced 256x128")
        ref = 0.5 * torch.stack([_ref(a[i], b[i % nb2], ta, tb) for i in range(batch)]) + res.double().cpu() + bias.double().cpu()[torch.arange(batch) % nb2][:, None, :]
        assert (outs[1].double().cpu() - ref).abs().max().item() < 2e-5 * K ** 0.5 * 4


def test_grouped_weight_gradients_match_single_launches_and_sum_the_bias(cuda):
    """dyn_gemm_f32_grouped: the deferred weight-gradient products of a backward pass as ONE launch over all their 128x128 tiles.
    Each group must equal the single-GEMM launch with the same tile and no K split bit for bit (same MFMA order), accumulate into
    dW with beta = 1, and the fused column sums of dy (the bias gradient) must match float64; K % 32 != 0 takes the register-staged
    path, as the short last window of a recording does (T' = 1992)."""
    from dynamic_asr_eval_amd import ops
    g = torch.Generator().manual_seed(5)
    for tokens in (2048, 1992, 100):
        shapes = [(3072, 768), (768, 3072), (2304, 768), (768, 768), (1536, 768), (256, 40), (132, 260)]
        descs, want, sums, outs, biases = [], [], [], [], []
        for n_out, k_in in shapes:
            dy = (torch.rand(tokens, n_out, generator=g) - 0.5).to(cuda)
            x = (torch.rand(tokens, k_in, generator=g) - 0.5).to(cuda)
            dw0 = (torch.rand(n_out, k_in, generator=g) - 0.5).to(cuda)
            b0 = (torch.rand(n_out, generator=g) - 0.5).to(cuda)
            single = dw0.clone()
            ops.gemm(dy, x, single, trans_a=True, M=n_out, N=k_in, K=tokens, lda=n_out, ldb=k_in, ldc=k_in, alpha=0.5, beta=1.0,
                     force=(128, 128, 1), split_k=1)
            want.append(single)
            sums.append(b0.double() + dy.double().sum(0))
            dw, b = dw0.clone(), b0.clone()
            outs.append(dw); biases.append(b)
            descs.append(ops.wgrad_desc(dy, x, dw, alpha=0.5, beta=1.0, colsum=b if n_out != 1536 else None, colsum_beta=1.0))
        ops.gemm_grouped(descs)
        for (n_out, k_in), dw, w, b, sref in zip(shapes, outs, want, biases, sums):
            assert torch.equal(dw, w), (tokens, n_out, k_in, (dw - w).abs().max().item())
            if n_out != 1536:
                assert (b.double() - sref).abs().max().item() < 2e-4, (tokens, n_out)


@pytest.mark.parametrize("M,N,K", [(4096, 3072, 768), (2048, 3072, 768), (2040, 2560, 256), (77, 100, 64)])
def test_silu_epilogues_equal_the_separate_kernels(cuda, M, N, K):
    """epilogue EPI_SILU (C = silu(v), aux = v) and EPI_SILU_GRAD (C = v * silu'(aux)) against GEMM + dyn_silu_fwd / dyn_silu_bwd,
    whatever plan (split-K, tail slices, edge tiles) the shape gets: the pre-activation is bit-identical, the activation agrees to
    the last bit or two (same formula; the compiler schedules the reciprocal differently inside the GEMM epilogue)."""
    def close(a, b):
        return (a - b).abs().max().item() <= 5e-7 * max(1.0, b.abs().max().item())

    from dynamic_asr_eval_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = (torch.rand(M, K, generator=g) - 0.5).to(cuda)
    w = (torch.rand(N, K, generator=g) - 0.5).to(cuda)
    bias = (torch.rand(N, generator=g) - 0.5).to(cuda)
    u_ref = ops.linear(x, w, bias)
    a_ref = ops.silu(u_ref)
    u = torch.empty_like(u_ref)
    a = ops.linear(x, w, bias, epilogue=ops.EPI_SILU, aux=u)
    assert torch.equal(u, u_ref) and close(a, a_ref)
    assert torch.equal(ops.linear(x, w, bias, epilogue=ops.EPI_SILU), a)          # no-grad form: the pre-activation is not kept
    dy = (torch.rand(M, N, generator=g) - 0.5).to(cuda)
    pre = (torch.rand(M, K, generator=g) * 6 - 3).to(cuda)
    d_ref = ops.silu_bwd(pre, ops.linear_dgrad(dy, w, alpha=0.5))
    d = ops.linear_dgrad(dy, w, alpha=0.5, epilogue=ops.EPI_SILU_GRAD, aux=pre)
    assert close(d, d_ref)


@pytest.mark.parametrize("mode,M,N,K,nb,force,split", [("NT", 4096, 2304, 768, 1, (128, 128, 4), 0), ("TN", 768, 3072, 2048, 1, (64, 64, 1), 3),
                                                       ("NN", 2048, 768, 3072, 1, (64, 64, 2), 0), ("NT", 1992, 1992, 128, 12, (128, 64, 8), 0),
                                                       ("TN", 256, 256, 20480, 1, (64, 64, 1), 12), ("NT", 1000, 900, 300, 2, (64, 128, 3), 0)])
def test_in_kernel_slice_combine_equals_the_reduce_pass(cuda, mode, M, N, K, nb, force, split):
    """K slices (global split-K or the K-sliced tail round) combined inside the kernel by the last-arriving workgroup (ordered
    ticket, slices summed in slice order) == slabs + the separate reduce kernel, bit for bit, with bias / beta / residual source;
    the arrival counters are back at zero afterwards, so the next launch (and a hipGraph replay) starts clean."""
    from dynamic_asr_eval_amd import ops
    ta, tb = mode[0] == "T", mode[1] == "T"
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.rand((nb, K, M) if ta else (nb, M, K), generator=g) - 0.5).to(cuda)
    b = (torch.rand((nb, N, K) if tb else (nb, K, N), generator=g) - 0.5).to(cuda)
    c0 = (torch.rand(nb, M, N, generator=g) - 0.5).to(cuda)
    res = (torch.rand(nb, M, N, generator=g) - 0.5).to(cuda)
    bias = (torch.rand(N, generator=g) - 0.5).to(cuda)

    def run(ticket):
        c = c0.clone()
        ops.gemm(a, b, c, trans_a=ta, trans_b=tb, M=M, N=N, K=K, lda=a.shape[2], ldb=b.shape[2], ldc=N, nb1=nb, sa=(a.shape[1] * a.shape[2], 0),
                 sb=(b.shape[1] * b.shape[2], 0), sc=(M * N, 0), alpha=0.7, beta=0.3, bias=bias, c_in=res, force=force, split_k=split, ticket=ticket)
        return c

    want = run(False)
    for _ in range(3):
        got = run(True)
        assert torch.equal(got, want), f"max |diff| {(got - want).abs().max().item()} at {int((got != want).sum())} elements"
    assert int(ops.counters(ops.workspace(cuda)).abs().sum().item()) == 0
    ref = 0.7 * ((a.transpose(1, 2) if ta else a).double() @ (b.transpose(1, 2) if tb else b).double()) + 0.3 * res.double() + bias.double()
    assert (want.double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
