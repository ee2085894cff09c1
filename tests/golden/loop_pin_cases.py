"""Cases and leaf stand-ins shared by tests/golden/make_loop_pins.py (runs the REFERENCE's loop functions on them, build container only)
and tests/test_reference_pins.py (runs oracle/*_ref.py on the same and compares with the stored outputs).  Nothing here reads
/root/reference."""
import argparse
import contextlib
import io
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import dynamic_eval_ref as R  # noqa: E402
from oracle.conformer_ref import SCConformerXLRef  # noqa: E402

TOY = dict(n_layers=2, d_model=64, n_heads=2, head_dim=32, subsampling_conv_channels=16)
VOCAB = 128


def content_masks(window, n_freq=3, freq_param=12):
    """SpecAugment stand-in: frequency masks that are a pure function of the window's CONTENT, so the reference function (which calls
    `augmentation(x)` without saying which window x is) and the oracle (which takes `fixed_masks[key]`) mask identically."""
    seed = int(window.double().abs().sum().item() * 1e3) % (2 ** 31 - 1)
    g = torch.Generator().manual_seed(seed)
    return (R.draw_masks(n_freq, freq_param, window.shape[-2], g), ([], []))


class StoredMaskSpecAugment:
    def __init__(self, **config):
        self.config = config

    def __call__(self, x):                      # [B, F, T]
        for b in range(x.shape[0]):
            R.apply_masks(x[b], content_masks(x[b]), self.config.get('zero_masking', False))
        return x


class OracleGreedyCTCDecoder:
    def __init__(self, tokenizer, blank_id):
        self.tokenizer, self.blank_id = tokenizer, blank_id

    def __call__(self, log_probs, decode=True):
        return self.tokenizer.decode(R.greedy_ctc_ids(log_probs, self.blank_id))


class _Inert:
    """plt / augment.EffectChain() / SoftDTW(...): every attribute is callable and returns the object itself."""
    def __getattr__(self, name):
        return self

    def __call__(self, *a, **k):
        return self


def tokenizer_128():
    import sentencepiece as spm
    return spm.SentencePieceProcessor(model_file=os.path.join(HERE, "tokenizer_128.model"))


def toy_model(seed, blank_bias=-0.3):
    m = SCConformerXLRef(TOY, vocab_size=VOCAB, seed=seed, blank_bias=blank_bias)
    m.device = torch.device("cpu")
    return m


def toy_args(**kw):
    a = argparse.Namespace(config={'model': {'subsampling_factor': 8}, 'audio_chunking': {'size': 512, 'overlap': 256}, 'training': {}})
    a.__dict__.update(kw)
    return a


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


DYNEVAL_CASES = {
    # tag: (spec frames, seq_len, overlap, args)
    "offline":   (1500, 512, 256, dict(optim_lr=3e-5, epochs=1)),
    "online":    (1500, 512, 256, dict(optim_lr=3e-5, epochs=1, online=True)),
    "epochs2":   (1100, 512, 256, dict(optim_lr=3e-5, epochs=2)),
    "online_e2": (1100, 512, 256, dict(optim_lr=3e-5, epochs=2, online=True)),       # lib.py:515 vs :527: the loop still runs 2 epochs
    "shuffle":   (1500, 512, 384, dict(optim_lr=3e-5, epochs=2, shuffle=True)),
    "short":     (300, 512, 256, dict(optim_lr=3e-5, epochs=1)),                    # seq_len > spec_n -> one window, overlap 0
    "config":    (1300, -1, -1, dict(optim_lr=3e-4, epochs=1, optim_weight_decay=0.01)),   # window / overlap from args.config
    "zero_mask": (1200, 512, 256, dict(optim_lr=3e-5, epochs=1, spec_augment_zero_masking=True)),
}
AWMC_CASES = {
    "e1": (1300, 512, 256, dict(optim_lr=3e-5, epochs=1)),
    "e2": (1100, 512, 384, dict(optim_lr=3e-5, epochs=2, ema_decay=0.9)),
}


CONCAT_CASES = (("plain", (700, 500, 300), dict(optim_lr=3e-5, epochs=1, seq_len=512, awmc=False), 256),
                ("e2_cfg", (600, 650), dict(optim_lr=2e-4, epochs=2, seq_len=-1, awmc=False), -1),
                ("awmc", (500, 600), dict(optim_lr=3e-5, epochs=1, seq_len=512, awmc=True), 256))
SU_CASES = (("e1", dict(epochs=1), 2e-5), ("e2_shuffle", dict(epochs=2, shuffle=True), 1e-5))


def params_digest(params):
    """What is stored of an updated parameter list: every 23rd element of the flattened list + its float64 sum and absolute sum."""
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    return flat[::23].numpy().copy(), [float(flat.double().sum()), float(flat.double().abs().sum())]


# ---- BASELINE config 5: the outer loop of lcasr/run_cross_dataset_eval.py on toy recordings
CROSS_CASES = (("same_overlap", (900, 700, 1100), (800, 600), dict(optim_lr=4e-6, epochs=1, seq_len=512, overlap=256, adapt_overlap=None,
                                                                       repeats=1, awmc=False, beamsearch=False, save_path='')),
               ("adapt_overlap", (1000, 650), (700, 900, 520), dict(optim_lr=3e-6, epochs=2, seq_len=512, overlap=256, adapt_overlap=384,
                                                                      repeats=1, awmc=False, beamsearch=False, save_path='')))


def toy_records(lens, seed):
    """Records with the harness contract {'process_fn': rec -> (spec [1, 80, T], gold text)} (reference run_cross_dataset_eval.py:106)."""
    recs = []
    for k, n in enumerate(lens):
        g = torch.Generator().manual_seed(seed + k)
        spec = torch.randn(1, 80, n, generator=g)
        words = ["onl", "k", "kl", "th", "on", "onl"]       # pieces the toy model actually emits: WERs below 1
        gold = " ".join(words[int(v)] for v in torch.randint(0, len(words), (max(2, n // 60),), generator=g))
        recs.append({'spec': spec, 'text': gold, 'process_fn': lambda r: (r['spec'], r['text'])})
    return recs
