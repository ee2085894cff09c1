"""End-to-end harness runs on the GPU with a tiny checkpoint and the small synthetic dataset: the reference's stdout
lines / pickle keys are produced (reference lcasr/run_dynamic_eval_full.py:117-148, run_cross_dataset_eval.py:200-218,
run_whole_concat_eval.py:157-183)."""
import argparse
import pickle

import pytest
import torch

pytestmark = pytest.mark.gpu
SMALL = dict(feat_in=80, n_layers=2, d_model=256, n_heads=2, head_dim=128, subsampling_factor=8, subsampling_conv_channels=64,
             conv_kernel_size=9, self_conditioning=True, rotary_base_freq=1500000)


def _ckpt(tmp_path, cuda, blank_bias=1.0):
    from dynamic_asr_eval_amd.model import SCConformerXL
    from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
    m = SCConformerXL(SMALL, vocab_size=128, device=cuda)
    init_synthetic(m, seed=1, blank_bias=blank_bias)
    path = str(tmp_path / "ckpt.pt")
    torch.save({'config': {'model': SMALL, 'audio_chunking': {'size': 16384, 'overlap': 0}, 'training': {'max_seq_len': 0}},
                'model': {k: v.cpu() for k, v in m.state_dict().items()}}, path)
    return path


def _argv(ckpt, extra):
    return ["-c", ckpt, "-seq", "512", "-o", "256", "-ds", "-nv", "-epochs", "1", "-kwargs", "optim_lr=1e-5", "vocab_size=128", "quiet=True",
            "spec_augment_n_freq_masks=2", "spec_augment_freq_mask_param=10"] + extra


def test_run_dynamic_eval_full(cuda, tmp_path, capsys):
    from dynamic_asr_eval_amd import lib, run_dynamic_eval_full as H
    ckpt = _ckpt(tmp_path, cuda)
    save = str(tmp_path / "res.pkl")
    args = lib.apply_args(H.build_parser(), ["-d", "synthetic_small", "-r", "2", "-s", save, "-log", str(tmp_path / "log.txt")] + _argv(ckpt, []))
    avg = H.main(args)
    out = capsys.readouterr().out
    assert "WER: " in out and "Average WER: " in out and avg >= 0
    for r in (1, 2):
        d = pickle.load(open(save.replace(".pkl", f"_{r}.pkl"), "rb"))
        assert set(d) >= {"wer", "words", "ins_rate", "del_rate", "sub_rate", "model_output", "gold", "elapsed_times", "args_dict", "repeat"}
        assert len(d["model_output"]) == 3 and d["repeat"] == f"{r}/2"
    assert "overlap: 256\t seq_len: 512\t WER:" in open(tmp_path / "log.txt").read()
    # several recordings in flight (chains=2): same transcripts as one at a time when the masks do not depend on draw order
    noaug = ["-c", ckpt, "-seq", "512", "-o", "256", "-ds", "-nv", "-epochs", "1", "-kwargs", "optim_lr=1e-5", "vocab_size=128", "quiet=True",
             "spec_augment_n_freq_masks=0"]
    res = {}
    for chains in (1, 2):
        sp = str(tmp_path / f"c{chains}.pkl")
        H.main(lib.apply_args(H.build_parser(), ["-d", "synthetic_small", "-s", sp] + noaug + [f"chains={chains}"]))
        res[chains] = pickle.load(open(sp.replace(".pkl", "_1.pkl"), "rb"))
    assert res[1]["model_output"] == res[2]["model_output"] and len(res[2]["elapsed_times"]) == 3
    # -kwargs lockstep=3: the three recordings (14 s, 9 s, 11.5 s: different lengths) advance through every window step in ONE batch
    sp = str(tmp_path / "lock.pkl")
    H.main(lib.apply_args(H.build_parser(), ["-d", "synthetic_small", "-s", sp] + noaug + ["lockstep=3"]))
    lock = pickle.load(open(sp.replace(".pkl", "_1.pkl"), "rb"))
    assert lock["model_output"] == res[1]["model_output"] and lock["wer"] == res[1]["wer"]
    # -awmc goes through the same harness (reference run_dynamic_eval_full.py:67-68)
    args = lib.apply_args(H.build_parser(), ["-d", "synthetic_small", "-awmc"] + _argv(ckpt, []))
    assert H.main(args) >= 0


def test_run_cross_dataset_and_whole_concat(cuda, tmp_path, capsys):
    from dynamic_asr_eval_amd import lib, run_cross_dataset_eval as X, run_whole_concat_eval as Wc
    ckpt = _ckpt(tmp_path, cuda)
    save = str(tmp_path / "x.pkl")
    args = lib.apply_args(X.build_parser(), ["-d", "synthetic_small", "-d2", "synthetic_small", "-split", "dev", "-s", save] + _argv(ckpt, []))
    X.main(args)
    d = pickle.load(open(save.replace(".pkl", "_1.pkl"), "rb"))
    assert set(d) >= {"a_baseline", "b_baseline", "a_to_b", "a_to_a_loo", "dataset_a", "dataset_b", "args_dict", "repeat"}
    assert len(d["a_to_b"]) == 2 and len(d["a_to_a_loo"]) == 2 and "wer" in d["a_to_b"][0]
    save2 = str(tmp_path / "w.pkl")
    args = lib.apply_args(Wc.build_parser(), ["-d", "synthetic_small", "-s", save2] + _argv(ckpt, []))
    Wc.main(args)
    out = capsys.readouterr().out
    assert "Baseline WER = " in out and "Adapted WER = " in out and "Delta = " in out
    d = pickle.load(open(save2.replace(".pkl", "_1.pkl"), "rb"))
    assert d["adapt_num_records"] == 3 and d["concat_total_frames"] == 1400 + 900 + 1150 and "delta_wer" in d


def test_run_seq_eval_outer_windows(cuda, tmp_path, capsys):
    """reference run_seq_eval.py: outer windows -> eval_fn -> outer stitch.  (1) one outer window covering the recording
    reproduces run_dynamic_eval_full's transcripts; (2) several outer windows in flight (chains=2) give the same transcripts
    as one at a time; (3) stdout / -log / pickle layout."""
    from dynamic_asr_eval_amd import lib, run_dynamic_eval_full as H, run_seq_eval as S
    ckpt = _ckpt(tmp_path, cuda)
    common = ["-c", ckpt, "-seq", "512", "-o", "256", "-ds", "-nv", "-epochs", "1", "-kwargs", "optim_lr=1e-5", "vocab_size=128", "quiet=True",
              "spec_augment_n_freq_masks=0", "min_minutes=0"]
    full = str(tmp_path / "full.pkl")
    H.main(lib.apply_args(H.build_parser(), ["-d", "synthetic_small", "-s", full] + common))
    whole = str(tmp_path / "whole.pkl")
    S.main(lib.apply_args(S.build_parser(), ["-d", "synthetic_small", "-s", whole, "-nsti_s", "-1"] + common))
    a = pickle.load(open(full.replace(".pkl", "_1.pkl"), "rb"))
    b = pickle.load(open(whole.replace(".pkl", "_1.pkl"), "rb"))
    # run_seq_eval takes test + dev (reference :59-61): the first three are the test split of run_dynamic_eval_full
    assert len(b["model_output"]) == 5 and b["model_output"][:3] == a["model_output"]
    outs = {}
    for chains in (1, 2):
        path = str(tmp_path / f"outer{chains}.pkl")
        avg = S.main(lib.apply_args(S.build_parser(), ["-d", "synthetic_small", "-s", path, "-nsti_s", "768", "-nsti_o", "256",
                                                       "-log", str(tmp_path / "log.txt")] + common + [f"chains={chains}"]))
        outs[chains] = pickle.load(open(path.replace(".pkl", "_1.pkl"), "rb"))
        assert avg >= 0
    assert outs[1]["model_output"] == outs[2]["model_output"] and len(outs[1]["model_output"]) == 5
    # the outer windows of a recording as a lockstep group (-kwargs lockstep=2): same transcripts
    path = str(tmp_path / "outer_lock.pkl")
    S.main(lib.apply_args(S.build_parser(), ["-d", "synthetic_small", "-s", path, "-nsti_s", "768", "-nsti_o", "256"] + common + ["lockstep=2"]))
    assert pickle.load(open(path.replace(".pkl", "_1.pkl"), "rb"))["model_output"] == outs[1]["model_output"]
    assert set(outs[1]) >= {"wer", "words", "ins_rate", "del_rate", "sub_rate", "model_output", "gold", "args_dict", "repeat"}
    out = capsys.readouterr().out
    assert "WER: " in out and "Average WER: " in out
    assert "overlap: 256\t seq_len: 512\t WER:" in open(tmp_path / "log.txt").read()


def test_outer_stitch_matches_host_rule(cuda):
    """The on-device outer stitch == the reference's host arithmetic (run_seq_eval.py:120-142) on random windows."""
    import numpy as np
    from dynamic_asr_eval_amd.run_seq_eval import outer_stitch
    g = torch.Generator().manual_seed(0)
    C, overlap = 17, 64
    wins = [(0, torch.log_softmax(torch.randn(48, C, generator=g), -1), 192), (128, torch.log_softmax(torch.randn(48, C, generator=g), -1), 192),
            (256, torch.log_softmax(torch.randn(31, C, generator=g), -1), 124)]
    got = outer_stitch([(k, lp.to(cuda), u) for k, lp, u in wins], overlap, C, cuda).cpu().numpy()
    total = sum(lp.shape[0] for _, lp, _ in wins)
    acc = np.zeros((total, C)); cnt = np.zeros((total, C)); pos = 0
    for k, lp, u in wins:
        ds = lp.shape[0]; ov = int(overlap / (u / ds))
        pos -= ov if k != 0 else 0
        cnt[pos:pos + ds] += 1; acc[pos:pos + ds] += np.exp(lp.numpy().astype(np.float64)); pos += ds
    keep = cnt.sum(-1) != 0
    ref = np.log(acc[keep] / cnt[keep])
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-5


def test_run_within_recording_loo_eval(cuda, tmp_path, capsys):
    """reference run_within_recording_loo_eval.py: audio-disjoint leave-one-out inside a recording; short recordings fall
    back to the windowed baseline; pickle layout of :218-228."""
    from dynamic_asr_eval_amd import lib, run_within_recording_loo_eval as L
    ckpt = _ckpt(tmp_path, cuda)
    save = str(tmp_path / "loo.pkl")
    args = lib.apply_args(L.build_parser(), ["-d", "synthetic_small", "-s", save, "-loo_s", "512", "-loo_o", "256"] + _argv(ckpt, []))
    L.main(args)
    d = pickle.load(open(save.replace(".pkl", "_1.pkl"), "rb"))
    assert set(d) >= {"loo", "baseline", "model_output", "baseline_model_output", "gold", "per_recording_meta", "dataset", "args_dict", "repeat"}
    assert [m["mode"] for m in d["per_recording_meta"]] == ["loo"] * 3 and all(m["n_chunks"] >= 3 for m in d["per_recording_meta"])
    out = capsys.readouterr().out
    assert "baseline WER" in out and "LOO WER" in out and "audio-disjoint LOO" in out
    # an outer chunk longer than the recording: fallback == the baseline transcript
    save2 = str(tmp_path / "loo2.pkl")
    L.main(lib.apply_args(L.build_parser(), ["-d", "synthetic_small", "-s", save2, "-loo_s", "65536", "-loo_o", "0", "-split", "dev"] + _argv(ckpt, ["chains=2"])))
    d2 = pickle.load(open(save2.replace(".pkl", "_1.pkl"), "rb"))
    assert all(m["mode"] == "fallback_windowed_eval" for m in d2["per_recording_meta"])
    assert d2["model_output"] == d2["baseline_model_output"]


def test_tedlium_and_chime6_shaped_adapters_through_the_harness(cuda, tmp_path):
    """process_fn of the TEDLIUM-shape (log-mel + ignored-segment zeroing) and CHiME-6-shape (channel average +
    renormalisation) adapters feeds eval_fn like the reference adapters do (tedlium/run.py:91-96, chime6/run.py:46-70)."""
    from dynamic_asr_eval_amd import datasets as D
    from dynamic_asr_eval_amd.datasets import proc_stm_lines
    from dynamic_asr_eval_amd.frontend import total_frames
    rec = D.get_text_and_audio_synthetic_tedlium('test', durations_s=[40.0])[0]
    spec, gold = rec['process_fn'](rec)
    assert spec.is_cuda and spec.shape == (1, 80, 4001) and len(gold) > 0 and 'ignore_time' not in gold
    _, keep, remove = proc_stm_lines(rec['stm'])
    for seg in remove:
        a, b = total_frames(seg['start']), min(total_frames(seg['end']), 4001)
        assert b > a and torch.all(spec[:, :, a:b] == 0)
    a, b = total_frames(keep[0]['start']), total_frames(keep[0]['end'])
    assert spec[:, :, a:b].abs().sum() > 0
    rec = D.get_text_and_audio_synthetic_chime6('dev', durations_s=[12.0], channels=3)[0]
    spec, gold = rec['process_fn'](rec)
    assert spec.shape == (1, 80, total_frames(10.0) - total_frames(1.5)) and spec.mean(-1).abs().max() < 1e-4
    # and through the harness (tiny model): both adapters are registered names
    from dynamic_asr_eval_amd import lib, run_dynamic_eval_full as H
    D.datasets_functions['tiny_ted'] = lambda split: D.get_text_and_audio_synthetic_tedlium(split, durations_s=[12.0, 9.0])
    ckpt = _ckpt(tmp_path, cuda)
    p = H.build_parser()
    for act in p._actions:
        if act.dest == 'dataset':
            act.choices = list(D.datasets_functions.keys())
    assert H.main(lib.apply_args(p, ["-d", "tiny_ted"] + _argv(ckpt, []))) >= 0


def test_run_in_dataset_eval_and_run_alias(cuda, tmp_path, capsys):
    """reference run_in_dataset_eval.py: adapt on recording 0, evaluate the rest with the adapted weights (epochs = 0);
    reference run.py is the run_dynamic_eval_full flow."""
    from dynamic_asr_eval_amd import lib, run as R0, run_dynamic_eval_full as H, run_in_dataset_eval as I
    assert R0.main is H.main
    ckpt = _ckpt(tmp_path, cuda)
    save = str(tmp_path / "ind.pkl")
    I.main(lib.apply_args(I.build_parser(), ["-d", "synthetic_small", "-s", save, "-ao", "0", "-log", str(tmp_path / "l.txt")] + _argv(ckpt, ["chains=2"])))
    d = pickle.load(open(save.replace(".pkl", "_1.pkl"), "rb"))
    assert set(d) >= {"wer", "words", "ins_rate", "del_rate", "sub_rate", "model_output", "gold", "args_dict", "repeat"}
    assert len(d["model_output"]) == 2                       # 3 test recordings: the first is only adapted on
    out = capsys.readouterr().out
    assert "Using adapt_overlap=0" in out and "WER: " in out and "Average WER: " in out


def test_bench_contract(cuda):
    """bench.py prints ONE JSON line with the driver's keys plus `roofline` (and `cpu_baseline` unless disabled)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--seconds", "200", "--steps", "2", "--warmup", "1", "--no_cpu_baseline", "--prewarm_s", "0"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"].startswith("audio-sec/s") and d["unit"] == "audio-s/s" and d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 2
    assert d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3 and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None or rf["traffic"] > 0
    # representative labels in the timed region, the degenerate (collapsing) labels and the boundary-faithful rate beside it
    assert d["config"]["label_tokens_per_window"] == 400 and d["value_degenerate_labels"] > 0 and d["value_boundary"] > 0 and d["value_online"] > d["value"]
    assert d["hyp_tokens_total"] >= 0 and "wer_counters" not in d
    # the path's other loops get a driver-visible number too (untimed for `value`)
    ow = d["other_workloads"]
    assert set(ow) == {"awmc", "wav2vec2_su", "enc_dec_teacher_ce", "soft_dtw"}, ow
    assert all(v.get("value", 0) > 0 and v["unit"] == "audio-s/s" for k, v in ow.items() if k != "soft_dtw"), ow
    assert len(ow["soft_dtw"]) == 4 and all(r["value_matches_oracle"] and r["us_per_diagonal"] > 0 for r in ow["soft_dtw"]), ow["soft_dtw"]
    assert 0 < ow["wav2vec2_su"]["roofline"]["frac"] < 1


def test_run_cross_speaker_gender(cuda, tmp_path, capsys):
    """reference run_cross_speaker_gender_tedlium.py with a 2 + 2 manifest over the TEDLIUM-shaped synthetic talks."""
    import json
    from dynamic_asr_eval_amd import datasets as D, lib, run_cross_speaker_gender_tedlium as G
    D.datasets_functions['tiny_ted4'] = lambda split: D.get_text_and_audio_synthetic_tedlium(
        split, durations_s=[9.0, 7.0] if split == 'test' else [8.0, 6.0])
    man = str(tmp_path / "m.json")
    json.dump({'name': 't', 'female': [{'talk_id': 'synthetic_tedlium_test_000'}, {'talk_id': 'synthetic_tedlium_dev_001'}],
               'male': [{'talk_id': 'synthetic_tedlium_test_001'}, {'talk_id': 'synthetic_tedlium_dev_000'}]}, open(man, 'w'))
    ckpt = _ckpt(tmp_path, cuda)
    p = G.build_parser()
    for act in p._actions:
        if act.dest == 'dataset':
            act.choices = list(D.datasets_functions.keys())
    save = str(tmp_path / "g.pkl")
    G.main(lib.apply_args(p, ["-d", "tiny_ted4", "--speaker_manifest", man, "-s", save] + _argv(ckpt, [])))
    d = pickle.load(open(save.replace(".pkl", "_1.pkl"), "rb"))
    assert set(d) >= {"male_baseline", "female_baseline", "male_to_male", "male_to_female", "female_to_female", "female_to_male", "args_dict", "repeat"}
    assert len(d["male_to_male"]) == 2 and len(d["female_to_male"]) == 2 and "wer" in d["male_to_female"][0]
    out = capsys.readouterr().out
    assert "Male baseline WER" in out and "Female baseline WER" in out and "2 female, 2 male" in out


def test_run_cross_dataset_matches_the_oracle_outer_loop(cuda, tmp_path, monkeypatch):
    """BASELINE config 5 above plumbing level (VERDICT r03 missing 1): the harness's whole flow — baselines over A and B at epochs 0, per i:
    adapt on A[i] with the ADAPT overlap and return_params, load the parameters, evaluate all of B and A minus {i}, restore — against
    oracle/cross_dataset_ref.py (the restatement of reference lcasr/run_cross_dataset_eval.py:92-218 that tests/golden/loop_pins pins to the
    reference's own statements) driving oracle.dynamic_eval_ref on the CPU with the same weights and stored SpecAugment masks.  Held: every
    scored corpus' transcripts in the reference's scoring order, and the four result entries (integer edit counters -> identical rates)."""
    import numpy as np
    from dynamic_asr_eval_amd import datasets as D, lib, run_cross_dataset_eval as X
    from dynamic_asr_eval_amd.tokenizer import SyntheticTokenizer
    from dynamic_asr_eval_amd.wer import basic_normalize
    from oracle import dynamic_eval_ref as R
    from oracle.conformer_ref import SCConformerXLRef
    from oracle.cross_dataset_ref import cross_dataset_ref, oracle_eval_fn
    from oracle.madgrad_ref import MADGRAD as MADGRAD_REF
    from oracle.wer_ref import word_error_rate_detail
    monkeypatch.setitem(D.datasets_functions, "toy_a", lambda split: D.get_text_and_audio_synthetic(split, durations_s=[9.0, 12.5, 7.0], seed=611))
    monkeypatch.setitem(D.datasets_functions, "toy_b", lambda split: D.get_text_and_audio_synthetic(split, durations_s=[8.0, 10.5], seed=733))
    ckpt = _ckpt(tmp_path, cuda, blank_bias=0.2)          # a low blank bias: transcripts of some tens of tokens per recording
    save = str(tmp_path / "x5.pkl")
    args = lib.apply_args(X.build_parser(), ["-d", "toy_a", "-d2", "toy_b", "-split", "dev", "-s", save, "-ao", "384", "-c", ckpt, "-seq", "512",
                                             "-o", "256", "-ds", "-nv", "-epochs", "1", "-kwargs", "optim_lr=2e-5", "vocab_size=128", "quiet=True"])
    g = torch.Generator().manual_seed(77)
    masks = {k: (R.draw_masks(2, 10, 80, g), ([], [])) for k in range(0, 2048, 128)}      # every window key of both overlaps
    args.spec_augment_fixed_masks = masks
    scored = []
    real_score = X.score_texts

    def recording_score(preds, golds, reduce_over_ranks=False):
        scored.append(list(preds))
        return real_score(preds, golds, reduce_over_ranks=reduce_over_ranks)
    monkeypatch.setattr(X, "score_texts", recording_score)
    got_lp = []
    real_transcribe = X.transcribe

    def recording_transcribe(decoder, logits):
        got_lp.append(logits.detach().float().cpu().numpy())
        return real_transcribe(decoder, logits)
    monkeypatch.setattr(X, "transcribe", recording_transcribe)
    X.main(args)
    got = pickle.load(open(save.replace(".pkl", "_1.pkl"), "rb"))

    state = torch.load(ckpt, map_location="cpu", weights_only=True)
    ref = SCConformerXLRef(dict(state["config"]["model"]), vocab_size=128)
    ref.load_state_dict(state["model"])
    ref.device = torch.device("cpu")
    tok = SyntheticTokenizer(128)
    oargs = argparse.Namespace(**{k: v for k, v in vars(args).items()})
    eval_fn = oracle_eval_fn(MADGRAD_REF, lib.get_lr_args_from_args, lambda a: {}, lambda spec, sl, ov: masks)
    want_scored, want_lp = [], []

    def oracle_transcribe(logits):
        want_lp.append(np.asarray(logits))
        return basic_normalize(tok.decode(R.greedy_ctc_ids(torch.as_tensor(logits), 128))).lower()
    want = cross_dataset_ref(oargs, ref, D.datasets_functions["toy_a"]("dev"), D.datasets_functions["toy_b"]("dev"), eval_fn, tok,
                             oracle_transcribe, word_error_rate_detail, record=want_scored, device_copies=True)[0]
    assert len(scored) == len(want_scored) == 2 + 2 * 3
    for k, (hyps, (phase, i, ref_hyps)) in enumerate(zip(scored, want_scored)):
        assert hyps == ref_hyps, f"corpus {k} ({phase}, i = {i}): transcripts differ from the oracle's"
    assert sum(len(h.split()) for hyps in scored for h in hyps) > 50, "the comparison must be over non-empty transcripts"
    # the seeded toy model's transcripts collapse to one token after ANY adaptation (MADGRAD's lr + eps floor moves every weight by ~1e-4), so
    # the order of the outer loop is held on the LOG-PROBS of every evaluated recording: 3 + 2 baselines, then per i: 2 of B and 2 of A \ {i}
    assert len(got_lp) == len(want_lp) == 5 + 3 * 4
    for k, (g_, w_) in enumerate(zip(got_lp, want_lp)):
        assert g_.shape == w_.shape and np.abs(g_ - w_).max() < 1e-3, f"evaluation {k}: log-probs differ from the oracle's by {np.abs(g_ - w_).max():.2e}"
    b0 = [want_lp[5 + 4 * i] for i in range(3)]           # B[0] after adapting on A[0], A[1], A[2]
    assert min(np.abs(b0[i] - b0[j]).max() for i in range(3) for j in range(i)) > 1e-2, "adapting on different A[i] must give different weights"
    assert np.abs(want_lp[3] - b0[0]).max() > 1e-2, "and they must differ from the unadapted baseline"
    for k in ("a_baseline", "b_baseline", "a_to_b", "a_to_a_loo"):
        assert got[k] == want[k], (k, got[k], want[k])
