"""wav2vec2 path: HIP Wav2Vec2ForCTC vs the transformers CPU model the reference loads
(reference wav2vec2/lib.py:20-23) with the same weights — logits, parameter gradients, and the per-utterance
dynamic-eval loop (reference wav2vec2/lib.py:293-462)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(cuda, seed=0):
    from transformers import Wav2Vec2Config, Wav2Vec2ForCTC as HF
    from dynamic_asr_eval_amd.wav2vec2_model import Wav2Vec2ForCTC
    torch.manual_seed(seed)
    cfg = Wav2Vec2Config(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, conv_dim=(256,) * 7,
                         num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, vocab_size=32, ctc_loss_reduction="mean")
    ref = HF(cfg).eval()
    with torch.no_grad():   # HF initialises biases / LN to trivial values: randomise so every gradient path is exercised
        for n, p in ref.named_parameters():
            if p.dim() == 1 or "original0" in n:
                p.add_(0.1 * torch.randn_like(p))
    hip = Wav2Vec2ForCTC(cfg, device=cuda)
    hip.load_state_dict(ref.state_dict(), strict=False)
    return ref, hip


def test_forward_backward_matches_transformers(cuda):
    ref, hip = _pair(cuda)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 6000, generator=g)
    out_ref = ref(x).logits
    out = hip(x.to(cuda)).logits
    assert out.shape == out_ref.shape
    err = (out.cpu() - out_ref).abs().max().item()
    assert err < 2e-4, err
    gl = torch.randn(out_ref.shape, generator=g) / out_ref.numel()
    out_ref.backward(gl)
    hip.zero_grad(); hip.backward(gl.to(cuda))
    grads = hip.grads_hf()
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert grads[n].abs().max().item() == 0.0, n       # masked_spec_embed is unused in eval mode
            continue
        # k_proj.bias has an analytically zero gradient (softmax is invariant to a constant key offset): absolute floor
        diff = (grads[n].cpu().reshape(p.grad.shape) - p.grad).abs().max().item()
        assert diff < 3e-3 * p.grad.abs().max().item() + 2e-8, (n, diff, p.grad.abs().max().item())
    # state_dict round trip keeps HF layouts
    sd = hip.state_dict()
    for n, p in ref.named_parameters():
        assert sd[n].shape == p.shape and torch.allclose(sd[n].cpu(), p.detach(), atol=0)


def test_active_subset_backward(cuda):
    ref, hip = _pair(cuda, seed=3)
    x = torch.randn(2, 5000, generator=torch.Generator().manual_seed(2)).to(cuda)
    out = hip(x).logits
    gl = torch.zeros_like(out); gl[0] = torch.randn(out.shape[1:], generator=torch.Generator().manual_seed(3)).to(cuda) / out[0].numel()
    hip.zero_grad(); hip.backward(gl); full = hip.flat_grads.clone()
    hip(x); hip.zero_grad(); hip.backward(gl[:1].contiguous(), n_active=1)
    assert (hip.flat_grads - full).abs().max().item() / full.abs().max().item() < 1e-5


def test_dynamic_eval_su_matches_oracle(cuda):
    """Per-utterance loop (reference wav2vec2/lib.py:293-462): normalise, forward B=2, greedy pseudo-label, CTC(mean),
    backward, clip_grad_norm_(10), MADGRAD step; utterance probs = log_softmax of the last copy."""
    import argparse
    from oracle.wav2vec2_ref import dynamic_eval_su_ref
    from oracle.madgrad_ref import MADGRAD as MADGRAD_REF
    from dynamic_asr_eval_amd import wav2vec2_lib as W
    ref, hip = _pair(cuda, seed=5)
    tok = W.CharTokenizer()
    g = torch.Generator().manual_seed(9)
    utts_ref = [{'waveform': torch.randn(1, n, generator=g) * 0.1 + 0.01} for n in (4000, 7000, 5200)]
    utts = [{'waveform': u['waveform'].clone()} for u in utts_ref]
    args = argparse.Namespace(epochs=1, shuffle=False)
    before = hip.flat_params.clone()
    dynamic_eval_su_ref(args, ref, utts_ref, tok, MADGRAD_REF, lr_args={'lr': 1e-5})
    W.dynamic_eval_su(args, hip, utts, 0, 0, tok, None, use_tqdm=False, optim=W.MADGRAD, lr_args={'lr': 1e-5})
    assert torch.equal(hip.flat_params, before)
    for a, b in zip(utts, utts_ref):
        assert a['probs'].shape == b['probs'].shape
        assert (a['probs'] - b['probs']).abs().max().item() < 1e-3
        assert torch.equal(a['probs'].argmax(-1), b['probs'].argmax(-1))
