"""Full-size checks (BASELINE.json configs[1] shapes: SCConformerXL 6 x 768, V+1 = 4096, 16384-frame windows, overlap
14336) through size-independent properties — the CPU oracle needs ~8 s per such window, so parity at this size is pinned by
invariants instead: normalised posteriors, exact weight restoration, eager == hipGraph replay == interleaved chains bit for
bit, stitched length and coverage, the zero-gradient shortcut, and GEMM linearity at the real shapes."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(**kw):
    a = argparse.Namespace()
    a.config = {'model': {'subsampling_factor': 8}, 'audio_chunking': {'size': 16384, 'overlap': 0}, 'training': {}}
    a.__dict__.update(dict(optim_lr=9e-5, epochs=1, shuffle=False, quiet=True, spec_augment_n_freq_masks=6,
                           spec_augment_freq_mask_param=34))
    a.__dict__.update(kw)
    return a


@pytest.fixture(scope="module")
def xl(cuda):
    from dynamic_asr_eval_amd.model import SCConformerXL
    from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    m = SCConformerXL(vocab_size=4095, device=cuda)
    init_synthetic(m, seed=0, blank_bias=1.0)
    return m, SyntheticTokenizer(4095)


def test_full_window_forward_is_a_distribution_and_backward_shortcut_is_exact(cuda, xl):
    model, _ = xl
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 80, 16384, generator=g).to(cuda)
    post = model(audio_signal=x)['final_posteriors']
    assert post.shape == (2, 2048, 4096) and torch.isfinite(post).all()
    assert (post.exp().sum(-1) - 1).abs().max().item() < 1e-4
    gp = torch.zeros_like(post)
    gp[0] = torch.randn(post.shape[1:], generator=g).to(cuda) / post[0].numel()
    model.zero_grad(); model.backward(gp)
    full = model.flat_grads.clone()
    model(audio_signal=x)
    model.zero_grad(); model.backward(gp[:1].contiguous(), n_active=1)
    assert torch.isfinite(full).all() and full.abs().max().item() > 0
    assert (model.flat_grads - full).abs().max().item() / full.abs().max().item() < 1e-5


def test_dynamic_eval_full_size_invariants(cuda, xl):
    from dynamic_asr_eval_amd import lib
    from dynamic_asr_eval_amd.model import SCConformerXL
    model, tok = xl
    before = model.flat_params.clone()
    from dynamic_asr_eval_amd.datasets import synthetic_spec
    spec = synthetic_spec(16384 + 2 * 2048 + 700, seed=3)                 # 3 full windows + one short tail window
    torch.manual_seed(11)
    eager = lib.dynamic_eval(_args(use_graphs=False), model, spec, 16384, 14336, tok, use_tqdm=False)
    assert torch.equal(model.flat_params, before), "weights must be restored bit for bit (reference lib.py:636-637)"
    # stitched output: one row per 8 input frames over the covered span, every row a distribution
    # windows at 0, 2048, 4096 and the 15036-frame tail at 6144: 2048-row outputs advance by 256 rows, the tail (1880 rows) lands at
    # 768 -> 2648 covered rows (reference lib.py:615-629 position rule)
    assert eager.shape == (2648, 4096)
    assert np.isfinite(eager).all() and np.abs(np.exp(eager.astype(np.float64)).sum(-1) - 1).max() < 1e-3
    # adaptation changes the result; epochs = 0 is the plain windowed forward
    plain = lib.dynamic_eval(_args(epochs=0), model, spec, 16384, 14336, tok, use_tqdm=False)
    assert plain.shape == eager.shape and np.abs(plain - eager).max() > 1e-6
    # hipGraph replay and two recordings in flight reproduce the eager sequential numbers exactly (same SpecAugment draws)
    torch.manual_seed(11)
    graphed = lib.dynamic_eval(_args(use_graphs=True), model, spec, 16384, 14336, tok, use_tqdm=False)
    assert np.array_equal(eager, graphed)
    twin = SCConformerXL(vocab_size=4095, device=cuda)
    twin.flat_params.copy_(model.flat_params)
    from oracle.dynamic_eval_ref import draw_masks, prepare_chunks
    _, keys = prepare_chunks(spec, 16384, 14336)
    mg = torch.Generator().manual_seed(5)
    masks = {k: (draw_masks(6, 34, 80, mg), ([], [])) for k in keys}
    one = lib.dynamic_eval(_args(spec_augment_fixed_masks=masks), model, spec, 16384, 14336, tok, use_tqdm=False)
    two = lib.dynamic_eval_many(_args(spec_augment_fixed_masks=masks), [model, twin], [spec, spec], 16384, 14336, tok, use_tqdm=False)
    assert np.array_equal(one, two[0]) and np.array_equal(one, two[1])
    assert torch.equal(model.flat_params, before) and torch.equal(twin.flat_params, before)


@pytest.mark.parametrize("mode,M,N,K", [("NT", 4096, 3072, 768), ("NN", 2048, 768, 3072), ("TN", 768, 3072, 2048), ("NT", 8192, 4096, 768)])
def test_gemm_linearity_at_real_shapes(cuda, mode, M, N, K):
    """A (x + y) = A x + A y and (2A) x = 2 (A x) to fp32 rounding at shapes of the adapt step."""
    from dynamic_asr_eval_amd import ops
    ta, tb = mode[0] == "T", mode[1] == "T"
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g).to(cuda)
    b1 = torch.randn((N, K) if tb else (K, N), generator=g).to(cuda)
    b2 = torch.randn((N, K) if tb else (K, N), generator=g).to(cuda)
    def mm(a_, b_):
        c = torch.empty(M, N, device=cuda)
        ops.gemm(a_, b_, c, trans_a=ta, trans_b=tb, M=M, N=N, K=K, lda=a_.shape[1], ldb=b_.shape[1], ldc=N)
        return c
    c1, c2, c12 = mm(a, b1), mm(a, b2), mm(a, b1 + b2)
    scale = (c1.abs().max() + c2.abs().max()).item()
    assert (c12 - (c1 + c2)).abs().max().item() < 1e-5 * scale * 4
    assert torch.equal(mm(2 * a, b1), 2 * c1)            # scaling by a power of two is exact in fp32


def test_fused_attention_in_the_no_grad_pass_matches_the_unfused_sequence(cuda, xl):
    """Final-pass batches of 4 full windows take the fused attention kernel (scores never materialised); the result agrees with
    the GEMM -> softmax -> GEMM sequence to fp32 rounding and decodes identically."""
    model, _ = xl
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 80, 16384, generator=g).to(cuda)
    with torch.no_grad():
        model.fused_attention = True
        a = model(audio_signal=x)['final_posteriors'].clone()
        model.fused_attention = False
        b = model(audio_signal=x)['final_posteriors'].clone()
        model.fused_attention = True
    assert (a - b).abs().max().item() < 1e-4 and not torch.equal(a, b)      # different code path (not bit-identical), same numbers
    assert torch.equal(a.argmax(-1), b.argmax(-1))


def test_config4_whole_concat_of_four_hours_at_full_size(cuda, xl):
    """BASELINE config 4 at its real size (reference lcasr/run_whole_concat_eval.py:123-152 -> run_half_concat_eval.adapt_on_concat_only):
    4 x 1 h of log-mel concatenated = 1 440 000 frames -> 697 windows of 16384 / 14336 (tests/golden/prepare_chunks.json, produced by
    the reference's own prepare_chunks), one adapt step per window on the 6 x 768 / V+1 = 4096 model, adapted weights returned and
    the model restored bit for bit; afterwards an epochs = 0 evaluation of one part with the adapted weights.  Properties only (the
    CPU oracle would need ~1.5 h for this): window count, finite / moved / restored weights, normalised posteriors, and the adapted
    weights actually changing the evaluation."""
    import json, os, time
    from dynamic_asr_eval_amd import lib
    from dynamic_asr_eval_amd.datasets import synthetic_spec
    from dynamic_asr_eval_amd.harness_common import restore_params, set_params
    from dynamic_asr_eval_amd.run_half_concat_eval import adapt_on_concat_only, concatenate_specs
    model, tok = xl
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prepare_chunks.json")))
    want = [c for c in gold["cases"] if c["spec_n"] == 1440000 and c["seq_len"] == 16384 and c["overlap"] == 14336][0]
    parts = [synthetic_spec(360000, seed=40 + i) for i in range(4)]
    concat = concatenate_specs(parts)
    assert concat.shape[-1] == 1440000
    _, keys = lib.prepare_chunks(concat, 16384, 14336)
    assert len(keys) == want["n_windows"] == 697
    before = model.flat_params.clone()
    args = _args(seq_len=16384, awmc=False, use_graphs=True)
    torch.cuda.synchronize(); t0 = time.time()
    updated = adapt_on_concat_only(args, model, concat, tok, adapt_overlap=14336)
    torch.cuda.synchronize(); dt = time.time() - t0
    assert torch.equal(model.flat_params, before), "weights must be restored after the adapt-only pass"
    flat_new = torch.cat([u.reshape(-1) for u in updated])
    assert torch.isfinite(flat_new).all()
    moved = sum(float((u - p.cpu()).abs().max()) > 0 for u, p in zip(updated, model.parameters()))
    assert moved > 0.9 * len(updated), f"only {moved} of {len(updated)} parameter tensors moved"
    part = parts[1][:, :, :16384 + 4 * 2048]
    base = lib.dynamic_eval(_args(epochs=0), model, part, 16384, 14336, tok, use_tqdm=False)
    set_params(model, updated)
    adapted = lib.dynamic_eval(_args(epochs=0), model, part, 16384, 14336, tok, use_tqdm=False)
    restore_params(model, before)
    assert torch.equal(model.flat_params, before)
    assert adapted.shape == base.shape and np.isfinite(adapted).all() and np.abs(np.exp(adapted.astype(np.float64)).sum(-1) - 1).max() < 1e-3
    assert np.abs(adapted - base).max() > 1e-6
    print(f"config 4: 697 adapt steps over 4 h of audio in {dt:.1f} s = {14400 / dt:.0f} audio-s/s (one chain, adapt only)")
