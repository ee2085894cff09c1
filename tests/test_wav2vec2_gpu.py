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


# ------------------------------------------------------------------------------------------------ base-960h shape + chunked loop + harness
def _base_pair(cuda, seed=0):
    """The architecture the reference loads (wav2vec2/lib.py:20-23, facebook/wav2vec2-base-960h = transformers' Wav2Vec2Config()
    defaults: 7 conv layers x 512 channels, positional conv k=128 g=16 with weight norm, 12 x 768, 12 heads, FFN 3072, vocab 32;
    94.4 M parameters) with seeded weights (no hub access offline)."""
    from transformers import Wav2Vec2Config, Wav2Vec2ForCTC as HF
    from dynamic_asr_eval_amd.wav2vec2_model import Wav2Vec2ForCTC
    torch.manual_seed(seed)
    cfg = Wav2Vec2Config()
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.conv_dim[0], cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups,
            cfg.vocab_size) == (768, 12, 512, 128, 16, 32)
    ref = HF(cfg).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.dim() == 1 or "original0" in n:
                p.add_(0.1 * torch.randn_like(p))
        ref.lm_head.bias[0] += 1.0     # favour the CTC blank (<pad> = 0): a seeded model otherwise labels nearly every frame and the
                                       # pseudo-label no longer fits the 409 frames (infinite CTC loss, in the reference too)
    hip = Wav2Vec2ForCTC(cfg, device=cuda)
    hip.load_state_dict(ref.state_dict(), strict=False)
    assert abs(sum(p.numel() for p in hip.parameters()) - sum(p.numel() for p in ref.parameters())) <= cfg.hidden_size   # masked_spec_embed
    return ref, hip


@pytest.fixture(scope="module")
def base_pair(cuda):
    return _base_pair(cuda)


def test_base_960h_shape_forward_and_every_gradient(cuda, base_pair):
    """Forward + every parameter gradient at the real architecture on 3 s and 2.3 s of audio (B = 2) vs the transformers CPU model."""
    ref, hip = base_pair
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 48000, generator=g)
    out_ref = ref(x).logits
    out = hip(x.to(cuda)).logits
    assert out.shape == out_ref.shape == (2, 149, 32)
    err = (out.cpu() - out_ref).abs().max().item()
    assert err < 5e-4, err
    assert torch.equal(out.cpu().argmax(-1), out_ref.argmax(-1)) or (out_ref.topk(2, -1).values.diff(dim=-1).abs().min().item() < 1e-4)
    gl = torch.randn(out_ref.shape, generator=g) / out_ref.numel()
    ref.zero_grad()
    out_ref.backward(gl)
    hip.zero_grad(); hip.backward(gl.to(cuda))
    grads = hip.grads_hf()
    worst = 0.0
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert grads[n].abs().max().item() == 0.0, n
            continue
        diff = (grads[n].cpu().reshape(p.grad.shape) - p.grad).abs().max().item()
        scale = p.grad.abs().max().item()
        worst = max(worst, diff / (scale + 1e-12)) if scale > 1e-7 else worst
        assert diff < 3e-3 * scale + 2e-8, (n, diff, scale)
    print("base-960h: forward err", err, "worst relative gradient error", worst)


@pytest.mark.parametrize("seq_len,overlap,epochs,L", [(6000, 0, 1, 15000), (6000, 1280, 1, 15000), (6000, 0, 2, 9000), (50000, 0, 1, 4000)])
def test_chunked_dynamic_eval_matches_oracle(cuda, seq_len, overlap, epochs, L):
    """Chunked loop (reference wav2vec2/lib.py:41-235): waveform windows by the inlined window rule, B = 2 clean copies, greedy
    pseudo-label, CTC(sum) / (N * B), MADGRAD step, exp(log_p[-1]) stitched with overlap_ds = int(overlap / (u_len / ds_len));
    several windows with a short tail, overlapping windows, 2 epochs (the last epoch's outputs are stitched), and a recording
    shorter than seq_len."""
    import argparse
    import numpy as np
    from oracle.wav2vec2_ref import dynamic_eval_chunked_ref
    from oracle.madgrad_ref import MADGRAD as MADGRAD_REF
    from dynamic_asr_eval_amd import wav2vec2_lib as W
    ref, hip = _pair(cuda, seed=7)
    tok = W.CharTokenizer()
    wav = torch.randn(1, L, generator=torch.Generator().manual_seed(L + overlap)) * 0.1 + 0.01
    args = argparse.Namespace(epochs=epochs, shuffle=False)
    before = hip.flat_params.clone()
    # r04: the first copy goes through the WavAugment chain of wav2vec2/lib.py:144-156 (100 x time_dropout(0.1 s) + the zero-noise
    # additive_noise = 0.5 x; draws from np.random as upstream): the same seed on both sides gives the same dropped spans
    np.random.seed(1000 + L)
    want = dynamic_eval_chunked_ref(args, ref, wav, seq_len, overlap, tok, MADGRAD_REF, lr_args={'lr': 1e-5})
    np.random.seed(1000 + L)
    got = W.dynamic_eval(args, hip, wav, seq_len, overlap, tok, None, use_tqdm=False, optim=W.MADGRAD, lr_args={'lr': 1e-5})
    assert torch.equal(hip.flat_params, before)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.abs(got - want).max() < 1e-3 and np.array_equal(got.argmax(-1), want.argmax(-1))
    if L == 15000 and overlap == 0:      # the augmentation is really applied: without it the result moves, and the device masking = the host's
        from oracle.wav2vec2_ref import wav_augment_ref
        np.random.seed(5)
        host = wav_augment_ref(wav[:, :6000].clone())
        np.random.seed(5)
        dev = W.wav_augment_chunk(wav[0, :6000].clone().to(cuda)).cpu()
        assert torch.equal(dev, host[0]) and int((host == 0).sum()) > 1000
        np.random.seed(1000 + L)
        clean = dynamic_eval_chunked_ref(args, ref, wav, seq_len, overlap, tok, MADGRAD_REF, lr_args={'lr': 1e-5}, wav_augment=False)
        assert np.abs(clean - want).max() > 1e-6


def test_chunked_dynamic_eval_at_the_reference_window(cuda, base_pair):
    """The reference's default window (`-seq 131072 -overlap 0`, wav2vec2/lib.py:480-481) at the base-960h architecture: a 9.4 s
    recording = one full 131072-sample window ([2, 131072] -> logits [2, 409, 32]) + a short tail window."""
    import argparse
    import numpy as np
    from oracle.wav2vec2_ref import dynamic_eval_chunked_ref
    from oracle.madgrad_ref import MADGRAD as MADGRAD_REF
    from dynamic_asr_eval_amd import wav2vec2_lib as W
    ref, hip = base_pair
    tok = W.CharTokenizer()
    wav = torch.randn(1, 150000, generator=torch.Generator().manual_seed(3)) * 0.1
    args = argparse.Namespace(epochs=1, shuffle=False)
    np.random.seed(77)
    want = dynamic_eval_chunked_ref(args, ref, wav, 131072, 0, tok, MADGRAD_REF, lr_args={'lr': 1e-6})
    np.random.seed(77)
    got = W.dynamic_eval(args, hip, wav, 131072, 0, tok, None, use_tqdm=False, optim=W.MADGRAD, lr_args={'lr': 1e-6})
    assert got.shape == want.shape == (409 + 58, 32) and np.isfinite(want).all(), "the oracle itself must produce a finite result"
    err = np.abs(got - want).max()
    bad = got.argmax(-1) != want.argmax(-1)
    top2 = np.sort(want, -1)[:, -2:]
    assert err < 1e-3 and not (bad & ((top2[:, 1] - top2[:, 0]) >= 5e-5)).any(), (err, int(bad.sum()))


def test_dynamic_eval_su_at_the_base_960h_shape(cuda, base_pair):
    """BASELINE config 3's named loop (`dynamic_eval_ctc_loss_su`, reference wav2vec2/lib.py:293-462) at the architecture the
    reference loads (Wav2Vec2Config() defaults, 94.4 M parameters): 4 utterances of 2-6 s cut out of one talk by the reference's own
    `fetch_utterances` rule (tedlium/run.py:56-83, pinned in tests/test_reference_pins.py), weights carried from utterance to
    utterance, against oracle/wav2vec2_ref.py on the transformers CPU model."""
    import argparse
    from oracle.wav2vec2_ref import dynamic_eval_su_ref
    from oracle.madgrad_ref import MADGRAD as MADGRAD_REF
    from dynamic_asr_eval_amd import wav2vec2_lib as W
    from dynamic_asr_eval_amd.datasets import fetch_utterances_from_lines
    ref, hip = base_pair
    tok = W.CharTokenizer()
    sr = 16000
    lines = ["talk 1 spk 0.00 2.10 <o> a b c", "talk 1 spk 2.10 8.00 <o> d e", "talk 1 spk 8.00 9.50 <o> ignore_time_segment_in_scoring",
             "talk 1 spk 9.50 13.25 <o> f", "talk 1 spk 13.25 17.40 <o> g h"]
    wave = torch.randn(1, int(17.4 * sr), generator=torch.Generator().manual_seed(21)) * 0.1 + 0.02
    utts, _ = fetch_utterances_from_lines(lines, wave, sr)
    assert [u['waveform'].shape[1] for u in utts] == [33600, 94400, 60000, 66400]
    utts_ref = [{'waveform': u['waveform'].clone()} for u in utts]
    args = argparse.Namespace(epochs=1, shuffle=False)
    before = hip.flat_params.clone()
    dynamic_eval_su_ref(args, ref, utts_ref, tok, MADGRAD_REF, lr_args={'lr': 1e-6})
    W.dynamic_eval_su(args, hip, utts, 0, 0, tok, None, use_tqdm=False, optim=W.MADGRAD, lr_args={'lr': 1e-6})
    assert torch.equal(hip.flat_params, before), "weights must be restored (reference wav2vec2/lib.py:455-456)"
    worst = 0.0
    for k, (a, b) in enumerate(zip(utts, utts_ref)):
        assert a['probs'].shape == b['probs'].shape and torch.isfinite(b['probs']).all()
        err = (a['probs'].cpu() - b['probs']).abs().max().item()
        worst = max(worst, err)
        assert err < 1e-3, f"utterance {k}: log-probs differ by {err}"
        bad = a['probs'].cpu().argmax(-1) != b['probs'].argmax(-1)
        top2 = b['probs'].topk(2, -1).values
        assert not (bad & ((top2[:, 0] - top2[:, 1]) >= 5e-5)).any(), f"utterance {k}: argmax differs away from a near-tie"
    print("dynamic_eval_su at base-960h: worst |dlogp| over 4 weight-carrying utterances", worst)


def test_run_wav2vec2_harness(cuda, tmp_path, capsys):
    """run_wav2vec2.py: both reference drivers' flow (tedlium/run.py -> dynamic_eval_su, earnings22/run.py -> dynamic_eval), the
    reference's flags, stdout lines and -log line; a local HF state_dict loads through -c and a foreign one is refused."""
    from transformers import Wav2Vec2Config, Wav2Vec2ForCTC as HF
    from dynamic_asr_eval_amd import run_wav2vec2 as R, wav2vec2_lib as W
    log = str(tmp_path / "log.txt")
    for mode in ("su", "chunked"):
        args = W.apply_args(R.build_parser(), ["--mode", mode, "--seconds", "20", "-seq", "131072", "-overlap", "0", "-nv", "-log", log, "-epochs", "1"])
        assert args.seq_len == 131072 and args.overlap == 0 and args.verbose is False and args.shuffle is False and args.split == "test"
        wer = R.main(args)
        out = capsys.readouterr().out
        assert "Loaded model from" in out and "Total number of parameters: 94." in out and f"WER: {wer}" in out
    lines = open(log).read().strip().split("\n")
    assert len(lines) == 2 and all("overlap: 0\t seq_len: 131072\t WER: " in l for l in lines)
    torch.manual_seed(0)
    ck = str(tmp_path / "hf.pt")
    torch.save(HF(Wav2Vec2Config()).state_dict(), ck)
    args = W.apply_args(R.build_parser(), ["--mode", "su", "--seconds", "6", "-c", ck, "-nv"])
    R.main(args)
    assert f"Loaded model from {ck}" in capsys.readouterr().out
    torch.save({"encoder.weight": torch.zeros(3)}, ck)
    with pytest.raises(KeyError):
        R.main(W.apply_args(R.build_parser(), ["--mode", "su", "--seconds", "6", "-c", ck, "-nv"]))


# ------------------------------------------------------------------------------------------------ r04: hipGraph replay over length buckets
def test_length_aware_kernels(cuda):
    """dyn_colnorm_fwd_len / _bwd_len, dyn_softmax_fwd_len, dyn_mask_rows: a DEVICE-side valid length gives, on the valid rows / columns, what
    the plain kernels give on the cut tensor; the length is read when the kernel runs (updating the tensor changes the next launch)."""
    from dynamic_asr_eval_amd import ops
    g = torch.Generator().manual_seed(4)
    B, T, C = 2, 300, 64
    x = torch.randn(B, T, C, generator=g).to(cuda)
    gamma, beta = torch.randn(C, generator=g).to(cuda), torch.randn(C, generator=g).to(cuda)
    dy = torch.randn(B, T, C, generator=g).to(cuda)
    valid = torch.zeros(1, dtype=torch.int32, device=cuda)
    for n in (300, 211, 64, 1):
        valid.fill_(n)
        y, mean, rstd = ops.colnorm(x, gamma, beta, 1e-5, valid=valid)
        yc, mc, rc = ops.colnorm(x[:, :n].contiguous(), gamma, beta, 1e-5)
        assert (mean - mc).abs().max().item() < 1e-6 and ((rstd - rc).abs() / rc).max().item() < 1e-5
        assert (y[:, :n] - yc).abs().max().item() < 1e-4 * max(1.0, yc.abs().max().item()) and torch.isfinite(y).all()
        dyz = dy.clone(); dyz[:, n:] = 0                          # what the padded frames carry in the model's backward
        dg, db = torch.zeros(C, device=cuda), torch.zeros(C, device=cuda)
        dgc, dbc = torch.zeros(C, device=cuda), torch.zeros(C, device=cuda)
        dx = ops.colnorm_bwd(x, gamma, mean, rstd, dyz, dg, db, valid=valid)
        dxc = ops.colnorm_bwd(x[:, :n].contiguous(), gamma, mc, rc, dy[:, :n].contiguous(), dgc, dbc)
        scale = max(1.0, dxc.abs().max().item())
        assert (dx[:, :n] - dxc).abs().max().item() < 2e-4 * scale and dx[:, n:].abs().max().item() == 0.0 if n < T else True
        assert (dg - dgc).abs().max().item() < 1e-3 * max(1.0, dgc.abs().max().item()) and (db - dbc).abs().max().item() < 1e-3 * max(1.0, dbc.abs().max().item())
        s = torch.randn(3, 5, T, T, generator=g).to(cuda)
        p = ops.softmax(s.clone(), valid=valid)
        pc = ops.softmax(s[..., :n].contiguous())
        assert torch.equal(p[..., :n], pc) and (n == T or p[..., n:].abs().max().item() == 0.0)      # bit for bit on the valid keys
        m = ops.mask_rows(x.clone(), valid)
        assert torch.equal(m[:, :n], x[:, :n]) and (n == T or m[:, n:].abs().max().item() == 0.0)


def test_bucketed_graph_replay_matches_the_unpadded_eager_run(cuda):
    """One captured launch sequence per length bucket (wav2vec2_model.py::forward): utterances of different lengths replay the SAME graphs with
    their frame counts in HBM.  Against the unpadded eager run of the same model: logits of the utterance's own frames within 2e-5, every
    parameter gradient within 1e-4 relative (same terms plus exact zeros, other summation order), also for a short utterance right after a long
    one (stale samples zeroed), a frozen prefix (another backward variant after the bucket's activations were released: recomputed eagerly), and
    with a byte budget that forces the buckets out and back in.  All buckets of a model share one memory pool."""
    ref, hip = _pair(cuda, seed=11)
    g = torch.Generator().manual_seed(12)
    hip.graph_after, hip.bucket_frames = 1, 32
    lengths = [9000, 10230, 6500, 9990, 3000, 10239, 6500]      # frames 27, 31, 20, 31, 9, 31, 20 -> buckets 32 (all)
    lengths += [10560, 20000, 12000]                            # 32 -> bucket 32 (exactly full), 62 -> 64, 37 -> 64
    def run(L, graphs, frozen=()):
        x = (torch.randn(2, L, generator=torch.Generator().manual_seed(L)) * 0.3).to(cuda)
        hip.use_graphs, hip.frozen = graphs, set(frozen)
        with torch.enable_grad():
            out = hip(x)
        assert hip._ctx_static == graphs
        T = out.frames
        logits = out.logits[:, :T].clone()
        gl = torch.zeros_like(out.logits[:1])
        gl[:, :T] = (torch.randn(1, T, logits.shape[-1], generator=torch.Generator().manual_seed(L + 1)) / T).to(cuda)   # zero past the utterance, as CTC gives
        hip.zero_grad(); hip.backward(gl.contiguous(), n_active=1)
        return T, logits, hip.flat_grads.clone()
    for budget in (96 << 30, 1):                                # 1 byte: every new bucket drops the others
        hip.graph_budget_bytes = budget
        if budget == 1:
            hip.drop_graphs()
            assert len(hip._graphs) == 0
        for L in lengths + ([9000] if budget == 1 else []):     # ... and the first bucket comes back in at the end
            fz = ("wav2vec2.feature_extractor",) if L == 9990 else ()
            T, lo, gr = run(L, True, fz)
            T2, lo2, gr2 = run(L, False, fz)
            assert T == T2 == hip.conv_lengths(L)[-1] and lo.shape == lo2.shape
            assert (lo - lo2).abs().max().item() < 2e-5 * max(1.0, lo2.abs().max().item()), (L, (lo - lo2).abs().max().item())
            assert (gr - gr2).abs().max().item() < 1e-4 * gr2.abs().max().item(), (L, (gr - gr2).abs().max().item(), gr2.abs().max().item())
            if fz:
                assert hip.G["wav2vec2.feature_extractor.conv_layers.0.conv.weight"].abs().max().item() == 0.0
        assert len(hip._graphs) == (2 if budget > 1 else 1)
    hip.use_graphs, hip.frozen = False, set()


def test_dynamic_eval_su_with_and_without_bucket_graphs(cuda):
    """The per-utterance loop with hipGraph replay over length buckets (default) against the same loop launched eagerly at every utterance's own
    length (`use_graphs=False`): 9 weight-carrying utterances in 3 buckets, log-probs within 2e-4, identical argmax away from near-ties."""
    import argparse
    from dynamic_asr_eval_amd import wav2vec2_lib as W
    ref, hip = _pair(cuda, seed=5)
    tok = W.CharTokenizer()
    g = torch.Generator().manual_seed(19)
    ns = (4000, 7000, 12000, 5200, 23000, 9000, 11000, 21000, 3100)
    a = [{'waveform': torch.randn(1, n, generator=g) * 0.1 + 0.01} for n in ns]
    b = [{'waveform': u['waveform'].clone()} for u in a]
    W.dynamic_eval_su(argparse.Namespace(epochs=1, shuffle=False), hip, a, 0, 0, tok, None, use_tqdm=False, optim=W.MADGRAD, lr_args={'lr': 1e-5})
    assert len(hip._graphs) >= 2 and not hip.use_graphs
    W.dynamic_eval_su(argparse.Namespace(epochs=1, shuffle=False, use_graphs=False), hip, b, 0, 0, tok, None, use_tqdm=False, optim=W.MADGRAD,
                      lr_args={'lr': 1e-5})
    for u, v, n in zip(a, b, ns):
        assert u['probs'].shape == v['probs'].shape == (hip.conv_lengths(n)[-1], 32)
        assert (u['probs'] - v['probs']).abs().max().item() < 2e-4
        bad = u['probs'].argmax(-1) != v['probs'].argmax(-1)
        top2 = v['probs'].topk(2, -1).values
        assert not (bad & ((top2[:, 0] - top2[:, 1]) >= 5e-5)).any()


def test_dynamic_eval_su_many_matches_one_talk_at_a_time(cuda):
    """wav2vec2_lib.dynamic_eval_su_many (talks in flight on their own streams and model replicas, one host thread) against dynamic_eval_su on every
    talk alone: with every utterance on the bucketed graph path (graph_after = 1) both run the same launch sequences — bit-identical log-probs;
    every replica's weights restored; a talk with no utterances and more talks than replicas."""
    import argparse
    from dynamic_asr_eval_amd import run_wav2vec2 as RW, wav2vec2_lib as W
    ref, hip = _pair(cuda, seed=5)
    hip.graph_after = 1
    tok = W.CharTokenizer()
    g = torch.Generator().manual_seed(23)
    sizes = [(4000, 7000, 12000, 5200), (9000, 3100, 23000), (), (6100, 6100, 15000, 4400, 8000)]
    talks = [[{'waveform': torch.randn(1, n, generator=g) * 0.1 + 0.01} for n in ns] for ns in sizes]
    args = argparse.Namespace(epochs=1, shuffle=False)
    want = [W.dynamic_eval_su(args, hip, [dict(u) for u in t], 0, 0, tok, None, use_tqdm=False, optim=W.MADGRAD, lr_args={'lr': 1e-5}) for t in talks]
    models = RW.replicate(hip, 2)
    assert models[1].graph_after == 1
    before = [m.flat_params.clone() for m in models]
    got = W.dynamic_eval_su_many(args, models, [[dict(u) for u in t] for t in talks], 0, 0, tok, None, optim=W.MADGRAD, lr_args={'lr': 1e-5})
    assert all(torch.equal(m.flat_params, b) for m, b in zip(models, before))
    assert len(got) == len(talks)
    for tw, tg, ns in zip(want, got, sizes):
        assert len(tw) == len(tg) == len(ns)
        for u, v in zip(tw, tg):
            assert u['probs'].shape == v['probs'].shape and torch.equal(u['probs'], v['probs']), (u['probs'] - v['probs']).abs().max().item()
    # the harness: 3 synthetic talks, 2 in flight
    a = W.apply_args(RW.build_parser(), ["--mode", "su", "--seconds", "12", "--talks", "3", "--chains", "2", "-nv"])
    assert isinstance(RW.main(a), float)
