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


def test_counter_based_dropout_and_sampling_kernels_match_their_restatement(cuda):
    """dyn_dropout / dyn_gumbel_argmax_rows (include/dyneval.h) against oracle/enc_dec_ref.py's numpy restatement of the same
    counter-based draws: masks bit-exact, sampled ids equal, and the sample frequencies follow softmax(x / T)."""
    from oracle.enc_dec_ref import dropout_mask, gumbel_argmax
    from dynamic_asr_eval_amd._lib import check, load
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(7, 333, generator=torch.Generator().manual_seed(1))
    for p, seed, stream in ((0.1, 0, 1 << 20), (0.5, 12345, (3 << 20) + 7), (0.0, 9, 2)):
        y = torch.empty_like(x, device=cuda)
        xd = x.to(cuda)
        check(load().dyn_dropout(xd.data_ptr(), y.data_ptr(), x.numel(), p, seed, stream, st), "dyn_dropout")
        assert torch.equal(y.cpu(), x * dropout_mask(x.shape, p, seed, stream)), (p, seed, stream)
    logits = torch.randn(50, 64, generator=torch.Generator().manual_seed(2)) * 2
    ids = torch.empty(50, dtype=torch.int32, device=cuda)
    ld = logits.to(cuda)
    check(load().dyn_gumbel_argmax_rows(ld.data_ptr(), 50, 64, 64, 1.0 / 0.7, 77, 1 << 20, ids.data_ptr(), st), "dyn_gumbel_argmax_rows")
    want = [gumbel_argmax(logits[r], 0.7, 77, (1 << 20) + r) for r in range(50)]
    assert ids.cpu().tolist() == want
    # distribution: 20000 draws of one row at T = 1.3
    row = torch.tensor([2.0, 1.0, 0.0, -1.0, 0.5, 1.5, -0.5, 0.2])
    many = row[None].repeat(20000, 1).contiguous().to(cuda)
    ids = torch.empty(20000, dtype=torch.int32, device=cuda)
    check(load().dyn_gumbel_argmax_rows(many.data_ptr(), 20000, 8, 8, 1.0 / 1.3, 5, 0, ids.data_ptr(), st), "dyn_gumbel_argmax_rows")
    freq = torch.bincount(ids.cpu().long(), minlength=8).float() / 20000
    assert (freq - torch.softmax(row / 1.3, -1)).abs().max().item() < 0.015, freq


def test_generate_with_kv_cache_greedy_and_sampled_matches_oracle(cuda):
    """`model.generate` (reference call sites lib.py:1128,1580,1620-1625): the incremental KV-cache decode equals the oracle's
    prefix re-run token for token — greedy, and sampled at temperature 0.7 / 1.5 with the shared counter-based draws; the host
    only looks for eos every `check_every` tokens, which must not change the result."""
    ref, hip = _pair(cuda, seed=5)
    x = torch.randn(1, 80, 480, generator=torch.Generator().manual_seed(8))
    want = ref.generate(x)["text_sequence"]
    for every in (1, 3, 8, 64):
        assert hip.generate(x.to(cuda), check_every=every)["text_sequence"] == want
    assert len(want) >= 3
    for temp, seed in ((0.7, 11), (1.5, 12)):
        w = ref.generate(x, sample=True, temperature=temp, seed=seed)["text_sequence"]
        g = hip.generate(x.to(cuda), sample=True, temperature=temp, seed=seed)["text_sequence"]
        assert g == w, (temp, g, w)
    assert ref.generate(x, sample=True, temperature=1.5, seed=12)["text_sequence"] != want     # the sampled decode really differs
    short = hip.generate(x.to(cuda), max_tokens=2)["text_sequence"]
    assert short == want[:2]


@pytest.mark.parametrize("dec", [dict(dec_d_model=256, dec_layers=2, dec_heads=4), dict(dec_d_model=512, dec_layers=3, dec_heads=4, dec_ff_mult=4),
                                 dict(dec_d_model=256, dec_layers=1, dec_heads=8, dec_ff_mult=3)])
def test_one_token_decoder_kernels_match_the_tile_kernel_path(cuda, dec):
    """dyn_decoder_steps (8 * layers + 2 lean launches per token: fused LayerNorm / embedding row-times-matrix kernel, one-query
    attention, argmax into the token buffer) against the same decode through dyn_gemm_f32 / softmax / norm launches at M = 1
    (`fused_decode = False`): the logits of EVERY step within 2e-5 of each other (both fp32, different summation order) when both
    paths are fed the same prefix, the greedy ids equal wherever the top-2 margin is not a rounding tie, the sampled ids equal (same
    counter-based draws), and the token ids independent of how many steps one call runs."""
    import ctypes
    from dynamic_asr_eval_amd import _lib
    from dynamic_asr_eval_amd.enc_dec import EncDecSCConformerXL
    from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
    cfg = dict(CFG, **dec)
    hip = EncDecSCConformerXL(cfg, vocab_size=VOCAB + 3, device=cuda)
    init_synthetic(hip, seed=21, blank_bias=1.0)
    assert hip.fused_decode
    x = torch.randn(1, 80, 640, generator=torch.Generator().manual_seed(9)).to(cuda)
    enc = hip.forward(x)
    n = 24
    got = {}
    for fused in (True, False):
        hip.fused_decode = fused
        got[fused] = [hip.generate(x, encoder_states=enc, max_tokens=n, check_every=e)["text_sequence"] for e in (1, 5, 64)]
        assert got[fused][0] == got[fused][1] == got[fused][2]
        hip._draws = 0
        got[fused].append(hip.generate(x, encoder_states=enc, sample=True, temperature=0.9, seed=4, max_tokens=n)["text_sequence"])
    # step-by-step logits on a common prefix (the tile-kernel path's own greedy ids)
    prefix = got[False][0]
    h = enc["hidden"][0]
    dd, L = hip.dec["dec_d_model"], hip.dec["dec_layers"]
    from dynamic_asr_eval_amd import ops
    from dynamic_asr_eval_amd.enc_dec import DEC
    with torch.no_grad(), ops.use_workspace(hip._scratch()):
        kv = [ops.linear(h, hip.P[f"{DEC}layers.{l}.cross.kv.weight"], hip.P[f"{DEC}layers.{l}.cross.kv.bias"]) for l in range(L)]
        caches = [[torch.zeros(len(prefix) + 2, 3 * dd, device=cuda) for _ in range(L)] for _ in range(2)]
        tok = torch.zeros(len(prefix) + 3, dtype=torch.int32, device=cuda)
        tok[1:len(prefix) + 1] = torch.tensor(prefix, dtype=torch.int32)
        tok2 = tok.clone()
        desc, keep = hip._decoder_desc(kv, caches[0], tok2, h.shape[0])
        worst, flips = 0.0, 0
        for t in range(len(prefix) + 1):
            want = hip._decoder_step(tok, t, h, kv, caches[1])[0].clone()
            _lib.check(_lib.load().dyn_decoder_steps(ctypes.byref(desc), t, 1, 0, 1.0, 0, 0, torch.cuda.current_stream().cuda_stream), "dyn_decoder_steps")
            have = keep[1].clone()
            worst = max(worst, float((have - want).abs().max()))
            top = want.topk(2).values
            if int(have.argmax()) != int(want.argmax()):
                assert float(top[0] - top[1]) < 1e-5
                flips += 1
            tok2[t + 1] = tok[t + 1]                          # keep both on the common prefix
        assert worst < 2e-5, worst
        assert flips <= 1
        for l in range(L):
            torch.testing.assert_close(caches[0][l][:len(prefix) + 1], caches[1][l][:len(prefix) + 1], rtol=0, atol=2e-5)
    if flips == 0:
        assert got[True][0] == got[False][0] and got[True][3] == got[False][3]
    assert len(got[False][0]) >= 3
    # argument checks fail loudly
    with pytest.raises(_lib.DynError):
        _lib.check(_lib.load().dyn_decoder_steps(ctypes.byref(desc), hip.dec["dec_max_positions"] - 1, 2, 0, 1.0, 0, 0, 0), "dyn_decoder_steps")
    with pytest.raises(_lib.DynError):
        _lib.check(_lib.load().dyn_decoder_steps(ctypes.byref(desc), 0, 1, 1, 0.0, 0, 0, 0), "dyn_decoder_steps")


def test_decoder_dropout_forward_and_every_gradient(cuda):
    """The three decoder dropout knobs (`dropout_emb`, `dropout_post_ff` -> ff_out_dropout, `dropout_attn` -> layer[0].fn.dropout_p;
    reference lcasr/lib.py:1511-1525,1636-1637) in training mode: loss and every gradient vs autograd with the same masks; in eval
    mode the knobs are inert."""
    from oracle.enc_dec_ref import _Streams, calc_loss_enc_dec_ref
    from dynamic_asr_eval_amd.enc_dec import calc_loss_enc_dec
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    ref, hip = _pair(cuda)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 80, 400, generator=g)
    text = torch.randint(1, VOCAB, (1, 11), generator=g)
    a_len, t_len = torch.LongTensor([400]), torch.LongTensor([11])
    tok = SyntheticTokenizer(VOCAB)
    plain = calc_loss_enc_dec(hip, x.to(cuda), text, a_len, t_len, tok, backward=False)["loss"]
    dec = ref.language_model_decoder
    dec.dropout_emb, dec.ff_out_dropout, dec.dropout_attn, dec.random_seed, dec.streams = 0.1, 0.2, 0.15, 42, _Streams()
    knobs = hip.language_model_decoder
    knobs.dropout_emb, knobs.ff_out_dropout = 0.1, 0.2
    for layer in knobs.layers:
        layer[0].fn.dropout_p = 0.15
    hip.random_seed, hip._draws = 42, 0
    assert calc_loss_enc_dec(hip, x.to(cuda), text, a_len, t_len, tok, backward=False)["loss"] == plain       # eval mode: inert
    assert hip._draws == 0
    dec.train(); knobs.train()
    loss_ref = calc_loss_enc_dec_ref(ref, x, text, a_len, t_len)
    loss_ref.backward()
    hip.zero_grad()
    out = calc_loss_enc_dec(hip, x.to(cuda), text, a_len, t_len, tok)
    dec.eval(); knobs.eval()
    assert hip._draws == dec.streams.draws == 1 + 2 * CFG["dec_layers"]
    assert abs(out["loss"] - float(loss_ref)) < 1e-4 * max(1.0, abs(float(loss_ref))) and abs(out["loss"] - plain) > 1e-3
    for (n, _), gh, p in zip(hip.named_parameters(), hip.grads(), ref.ordered_parameters()):
        rel = (gh.cpu() - p.grad).abs().max().item() / (p.grad.abs().max().item() + 1e-12)
        assert rel < 2e-3, (n, rel)


def test_enc_dec_dynamic_eval_with_decode_agreement_and_dropout_matches_oracle(cuda):
    """The loop with `teacher_filter_decode_agreement` (a second, sampled decode per window, lib.py:1620-1627; the CER-similarity
    decision of enc_dec_teacher_filters.py is pinned to the reference) and the three dropout knobs on: same teacher / agreement
    texts, same skip decisions, same adapted parameters and final transcript as the oracle loop."""
    from oracle import dynamic_eval_ref as R
    from oracle.enc_dec_ref import enc_dec_dynamic_eval_ref
    from oracle.madgrad_ref import MADGRAD as MADGRAD_REF
    from dynamic_asr_eval_amd.enc_dec import enc_dec_dynamic_eval
    from dynamic_asr_eval_amd.enc_dec_teacher_filters import add_enc_dec_teacher_filter_args, should_skip_faulty_teacher_prediction
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    ref, hip = _pair(cuda, seed=7)
    tok = SyntheticTokenizer(VOCAB)
    spec = torch.randn(1, 80, 1000, generator=torch.Generator().manual_seed(4))
    _, keys = R.prepare_chunks(spec, 256, 0)
    mg = torch.Generator().manual_seed(6)
    masks = {k: (R.draw_masks(3, 12, 80, mg), ([], [])) for k in keys}
    defaults = vars(add_enc_dec_teacher_filter_args(argparse.ArgumentParser()).parse_args([]))
    args = argparse.Namespace(**dict(defaults, teacher_filter_decode_agreement=True, teacher_decode_agreement_temperature=0.6,
                                     teacher_decode_agreement_min_similarity=0.65))
    args.__dict__.update(dict(config={'model': {'subsampling_factor': 8}, 'audio_chunking': {'size': 2048, 'overlap': 0}, 'training': {}},
                              optim_lr=1e-4, epochs=1, shuffle=False, training_mode='teacher_ce', spec_augment_fixed_masks=masks,
                              dropout_emb=0.1, dropout_post_ff=0.1, dropout_attn=0.1, random_seed=3))
    decisions, trace = [], []

    def skip_fn(tokens, text, frames, agreement_text):
        s, _ = should_skip_faulty_teacher_prediction(args=args, teacher_pred_tokens=tokens, teacher_pred_text=text, spec_frames=frames,
                                                     agreement_text=agreement_text)
        decisions.append(s)
        return s

    want, p_ref = enc_dec_dynamic_eval_ref(ref, spec, 256, tok, MADGRAD_REF, {'lr': 1e-4}, epochs=1, fixed_masks=masks, return_params=True,
                                           skip_fn=skip_fn, dropout_emb=0.1, dropout_post_ff=0.1, dropout_attn=0.1, agreement_temperature=0.6,
                                           random_seed=3, trace=trace)
    assert any(a is not None and a != t for t, a in trace), "the sampled decode should differ from the greedy teacher somewhere"
    before = hip.flat_params.clone()
    got, p = enc_dec_dynamic_eval(args, hip, spec, 256, 0, tok, use_tqdm=False, return_params=True)
    assert torch.equal(hip.flat_params, before) and got == want
    assert any(decisions) and not all(decisions), f"seeded case: the first window is rejected (1 - CER 0.55 < 0.65), the others train: {decisions}"
    for a, b in zip(p, p_ref):
        assert (a - b).abs().max().item() < 5e-5
    assert hip.language_model_decoder.training is False and all(l[0].fn.dropout_p == 0 for l in hip.language_model_decoder.layers)


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
