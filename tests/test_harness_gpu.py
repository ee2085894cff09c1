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


def _ckpt(tmp_path, cuda):
    from dynamic_asr_eval_amd.model import SCConformerXL
    from dynamic_asr_eval_amd.synthetic_weights import init_synthetic
    m = SCConformerXL(SMALL, vocab_size=128, device=cuda)
    init_synthetic(m, seed=1, blank_bias=1.0)
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
