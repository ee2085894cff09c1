"""Encoder-decoder `teacher_ce` adaptation (reference lcasr/lib.py:1228-1322,1475-1732 and the drivers
enc_dec_dynamic_eval_test.py / enc_dec_inference_test.py) on the HIP path against oracle/enc_dec_ref.py (torch CPU + autograd) with
shared seeded weights.  The decoder architecture is builder-defined on both sides (parity unpinned against upstream)."""
import argparse
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CFG = dict(n_layers=2, d_model=256, n_heads=2, head_dim=128, subsampling_conv_channels=64, dec_d_model=256, dec_layers=2, dec_heads=4,
           ctc_loss_weight=0.3)
VOCAB = 64


def _pair(cuda, seed=3):
    from oracle.enc_dec_ref import EncDecRef
    from dynamic_asr_eval_amd.enc_dec import EncDecSCConformerXL
    ref = EncDecRef(CFG, vocab_size=VOCAB, seed=seed, blank_bias=1.0)
    hip = EncDecSCConformerXL(CFG, vocab_size=VOCAB, device=cuda)
    sd = ref.hip_state_dict()
    assert sorted(sd) == sorted(n for n, _ in hip.spec)
    hip.load_state_dict(sd)
    return ref, hip


def test_decoder_forward_loss_and_every_gradient(cuda):
    """calc_loss_enc_dec (lib.py:1228-1322): teacher-forced decoder logits, the mixed CTC + LM loss and the gradient of every encoder
    and decoder parameter (the cross-attention's gradient flows back into the encoder states) vs autograd."""
    from oracle.enc_dec_ref import calc_loss_enc_dec_ref
    from dynamic_asr_eval_amd.enc_dec import calc_loss_enc_dec
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    ref, hip = _pair(cuda)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 80, 400, generator=g)
    text = torch.randint(1, VOCAB, (1, 9), generator=g)
    a_len, t_len = torch.LongTensor([400]), torch.LongTensor([9])
    loss_ref = calc_loss_enc_dec_ref(ref, x, text, a_len, t_len)
    loss_ref.backward()
    with torch.no_grad():
        out_ref = ref.forward(x, torch.nn.functional.pad(text, (1, 0), value=0), a_len)
    hip.zero_grad()
    out = calc_loss_enc_dec(hip, x.to(cuda), text, a_len, t_len, SyntheticTokenizer(VOCAB))
    assert (out["lm_posteriors"].cpu() - out_ref["final_posteriors_lm"]).abs().max().item() < 2e-4
    assert (out["ctc_posteriors"].cpu() - out_ref["final_posteriors_ctc"]).abs().max().item() < 2e-4
    assert abs(out["loss"] - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref)))
    worst = 0.0
    for (n, _), gh, p in zip(hip.named_parameters(), hip.grads(), ref.ordered_parameters()):
        rel = (gh.cpu() - p.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-12)
        worst = max(worst, rel)
        assert rel < 2e-3, (n, rel)
    print("enc-dec worst relative gradient error", worst)


def test_generate_and_inference_match_oracle(cuda):
    from oracle.enc_dec_ref import enc_dec_inference_ref
    from dynamic_asr_eval_amd.enc_dec import enc_dec_inference
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    ref, hip = _pair(cuda, seed=5)
    tok = SyntheticTokenizer(VOCAB)
    spec = torch.randn(1, 80, 700, generator=torch.Generator().manual_seed(2))
    want = enc_dec_inference_ref(ref, spec, 256, 0, tok)
    got = enc_dec_inference(hip, spec, 256, 0, tok, use_tqdm=False)
    assert got == want and len(got.split()) >= 3


@pytest.mark.parametrize("filters", [{}, {"teacher_filter_max_length": True, "teacher_min_frames_per_token": 24}])
def test_enc_dec_dynamic_eval_teacher_ce_matches_oracle(cuda, filters):
    """The loop (lib.py:1475-1732, teacher_ce): greedy teacher on the clean copy, optional filters, one supervised MADGRAD step per
    window on the augmented copy, final decode with the adapted weights, weights restored."""
    from oracle import dynamic_eval_ref as R
    from oracle.enc_dec_ref import enc_dec_dynamic_eval_ref
    from oracle.madgrad_ref import MADGRAD as MADGRAD_REF
    from dynamic_asr_eval_amd.enc_dec import enc_dec_dynamic_eval
    from dynamic_asr_eval_amd.enc_dec_teacher_filters import add_enc_dec_teacher_filter_args, should_skip_faulty_teacher_prediction
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    ref, hip = _pair(cuda, seed=7)
    tok = SyntheticTokenizer(VOCAB)
    spec = torch.randn(1, 80, 700, generator=torch.Generator().manual_seed(4))
    _, keys = R.prepare_chunks(spec, 256, 0)
    mg = torch.Generator().manual_seed(6)
    masks = {k: (R.draw_masks(3, 12, 80, mg), ([], [])) for k in keys}
    defaults = vars(add_enc_dec_teacher_filter_args(argparse.ArgumentParser()).parse_args([]))
    args = argparse.Namespace(**dict(defaults, **filters))
    args.__dict__.update(dict(config={'model': {'subsampling_factor': 8}, 'audio_chunking': {'size': 2048, 'overlap': 0}, 'training': {}},
                              optim_lr=1e-4, epochs=1, shuffle=False, training_mode='teacher_ce', spec_augment_fixed_masks=masks))
    skipped = []

    def skip_fn(tokens, text, frames):
        s, _ = should_skip_faulty_teacher_prediction(args=args, teacher_pred_tokens=tokens, teacher_pred_text=text, spec_frames=frames)
        skipped.append(s)
        return s

    want, p_ref = enc_dec_dynamic_eval_ref(ref, spec, 256, tok, MADGRAD_REF, {'lr': 1e-4}, epochs=1, fixed_masks=masks, return_params=True, skip_fn=skip_fn)
    before = hip.flat_params.clone()
    got, p = enc_dec_dynamic_eval(args, hip, spec, 256, 0, tok, use_tqdm=False, return_params=True)
    assert torch.equal(hip.flat_params, before)
    assert got == want
    if filters:
        assert any(skipped)
    for a, b in zip(p, p_ref):
        assert (a - b).abs().max().item() < 5e-5
    with pytest.raises(NotImplementedError):
        args.training_mode = 'grpo'
        enc_dec_dynamic_eval(args, hip, spec, 256, 0, tok, use_tqdm=False)


def test_enc_dec_harnesses(cuda, tmp_path, capsys):
    """enc_dec_dynamic_eval_test.py / enc_dec_inference_test.py mirrors: flags, stdout lines, -log line, pickle keys."""
    from dynamic_asr_eval_amd import enc_dec_dynamic_eval_test as A, enc_dec_inference_test as I, lib
    from dynamic_asr_eval_amd.enc_dec import EncDecSCConformerXL
    from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
    m = EncDecSCConformerXL(CFG, vocab_size=128, device=cuda)
    init_synthetic(m, seed=1, blank_bias=1.0)
    ck = str(tmp_path / "encdec.pt")
    model_cfg = dict(CFG, feat_in=80, subsampling_factor=8, conv_kernel_size=9, self_conditioning=True, rotary_base_freq=1500000)
    torch.save({'config': {'model': model_cfg, 'audio_chunking': {'size': 2048, 'overlap': 0}, 'training': {'max_seq_len': 0}},
                'model': {k: v.cpu() for k, v in m.state_dict().items()}}, ck)
    common = ["-c", ck, "-seq", "512", "-o", "0", "-nv", "-kwargs", "optim_lr=1e-5", "vocab_size=128", "spec_augment_n_freq_masks=2", "spec_augment_freq_mask_param=10"]
    save, log = str(tmp_path / "a.pkl"), str(tmp_path / "log.txt")
    avg = A.main(lib.apply_args(A.build_parser(), ["-d", "synthetic_small", "-s", save, "-log", log, "--training_mode", "teacher_ce", "--breaks",
                                                   "--teacher_filter_max_length"] + common))
    out = capsys.readouterr().out
    assert "WER: " in out and "Average WER: " in out and "Teacher pred:" in out and "Saved to" in out and avg >= 0
    d = pickle.load(open(save.replace(".pkl", "_1.pkl"), "rb"))
    assert set(d) >= {"wer", "words", "ins_rate", "del_rate", "sub_rate", "model_output", "gold", "elapsed_times", "args_dict", "repeat"}
    assert len(d["model_output"]) == 1 and d["repeat"] == "1/1"
    save2 = str(tmp_path / "i.pkl")
    wer = I.main(lib.apply_args(I.build_parser(), ["-d", "synthetic_small", "-s", save2, "-log", log, "--ctc_greedy"] + common))
    out = capsys.readouterr().out
    assert f"WER: {wer}" in out and "Generated text:" in out and "CTC greedy:" in out
    d2 = pickle.load(open(save2.replace(".pkl", "_1.pkl"), "rb"))
    assert len(d2["model_output"]) == 3 and d2["repeat"] == "1/1"
    assert open(log).read().count("overlap: 0\t seq_len: 512\t WER: ") == 2
    with pytest.raises(NotImplementedError):
        A.main(lib.apply_args(A.build_parser(), ["-d", "synthetic_small", "--breaks"] + common))      # default training_mode grpo: RL, out of scope
