"""HIP kernels (through the C-ABI) against the COMMITTED fixtures of tests/golden/ — no oracle call at run time.
The fixtures come from the ops the reference itself calls (torch.nn.CTCLoss, torch.optim.Adam) or from the oracle's
restatement of its loop; tests/golden/make_golden.py is the generator, tests/test_host_cpu.py checks that the oracle still
reproduces them."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ctc_loss_and_grad_match_the_torch_fixture(cuda):
    """torch.nn.CTCLoss(blank=128, reduction='sum') — reference lcasr/lib.py:492,575,579."""
    from dynamic_asr_eval_amd import ops
    d = np.load(os.path.join(GOLDEN, "ctc_64x10x129.npz"))
    lp = torch.from_numpy(d["log_probs"]).transpose(0, 1).contiguous().to(cuda)          # [T, B, C] -> [B, T, C]
    loss, nll, grad = ops.ctc_loss(lp, torch.from_numpy(d["targets"]).to(cuda), torch.from_numpy(d["input_lengths"]).to(cuda),
                                   torch.from_numpy(d["target_lengths"]).to(cuda), 128, reduction="sum")
    assert abs(loss.item() - float(d["loss"])) < 1e-5 * abs(float(d["loss"])), (loss.item(), float(d["loss"]))
    ref = torch.from_numpy(d["grad"]).transpose(0, 1)
    err = (grad.cpu() - ref).abs().max().item()
    assert err < 1e-4, err          # fp32 log-space lattice over T = 64 steps; |grad| <= 1
    # frames past input_lengths carry no gradient (torch zero-fills them)
    assert torch.all(grad[1, int(d["input_lengths"][1]):] == 0)


def test_adam_and_madgrad_trajectories_match_the_fixtures(cuda):
    """torch.optim.Adam (reference nvidia_ctc/lib.py:43) and the MADGRAD restatement (lcasr/lib.py:458)."""
    from dynamic_asr_eval_amd import ops
    d = np.load(os.path.join(GOLDEN, "adam_3step.npz"))
    p = torch.from_numpy(d["p0"]).to(cuda); m = torch.empty_like(p); v = torch.empty_like(p)
    for k in range(3):
        ops.adam_step(p, torch.from_numpy(d["grads"][k]).to(cuda), m, v, float(d["lr"]), 0.9, 0.999, 1e-8, 0.0, k)
        assert np.abs(p.cpu().numpy() - d["params"][k]).max() < 2e-6
    d = np.load(os.path.join(GOLDEN, "madgrad_3step.npz"))
    p = torch.from_numpy(d["p0"]).to(cuda); s = torch.empty_like(p); nu = torch.empty_like(p); x0 = torch.empty_like(p)
    for k in range(3):
        ops.madgrad_step(p, torch.from_numpy(d["grads"][k]).to(cuda), s, nu, x0, float(d["lr"]), 0.9, 0.0, 1e-6, k)
        assert np.abs(p.cpu().numpy() - d["params"][k]).max() < 5e-6


def test_dynamic_eval_reproduces_the_committed_trace(cuda):
    """End to end: seeded 2-layer model, 1100-frame recording, 512/256 windows, stored SpecAugment masks -> the stitched
    log-probs, CTC argmax ids (bit-exact bar) and adapted-parameter checksum recorded by the oracle's loop."""
    from oracle.conformer_ref import SCConformerXLRef       # only to rebuild the SAME seeded weights (no oracle compute)
    from dynamic_asr_eval_amd import lib
    from dynamic_asr_eval_amd.model import SCConformerXL
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    d = np.load(os.path.join(GOLDEN, "dyneval_trace.npz"))
    cfg = dict(n_layers=2, d_model=256, n_heads=2, head_dim=128, subsampling_conv_channels=64)
    ref = SCConformerXLRef(cfg, vocab_size=128, seed=int(d["model_seed"]), blank_bias=float(d["blank_bias"]))
    hip = SCConformerXL(cfg, vocab_size=128, device=cuda)
    hip.load_state_dict(ref.state_dict())
    masks = {int(k): ((list(map(int, s)), list(map(int, w))), ([], [])) for k, s, w in zip(d["keys"], d["mask_starts"], d["mask_widths"])}
    a = argparse.Namespace(config={'model': {'subsampling_factor': 8}, 'audio_chunking': {'size': 16384, 'overlap': 0}, 'training': {}},
                           optim_lr=float(d["lr"]), epochs=1, shuffle=False, online=False, spec_augment_fixed_masks=masks, quiet=True)
    out, params = lib.dynamic_eval(a, hip, torch.from_numpy(d["spec"]), 512, 256, SyntheticTokenizer(128), use_tqdm=False, return_params=True)
    assert out.shape == d["logits"].shape
    assert np.abs(out - d["logits"]).max() < 1e-3
    assert np.array_equal(out.argmax(-1).astype(np.int32), d["argmax"])
    cs = np.array([float(sum(p.double().sum() for p in params)), float(sum(p.double().abs().sum() for p in params))])
    assert np.abs(cs - d["param_checksum"]).max() < 1e-2 * 1e-1 + 1e-6 * abs(d["param_checksum"][1])
