"""Toy wav2vec2 fixtures shared by tests/golden/make_loop_pins.py (which runs the REFERENCE's per-utterance loop on them) and
tests/test_reference_pins.py (which runs oracle/wav2vec2_ref.py on the same): a tiny Wav2Vec2ForCTC config, the 32-symbol character
tokenizer surface of wav2vec2-base-960h (`blank_id`, `vocab`, `decode`, `tokenizer(text).input_ids`; the HF tokenizer needs downloaded
files) and a processor whose `feature_extractor` is the REAL transformers Wav2Vec2FeatureExtractor (default arguments need no files)."""
from types import SimpleNamespace

import torch

W2V2_TOY = dict(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, conv_dim=(32,) * 7,
                num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, vocab_size=32, ctc_loss_reduction="mean")


class CharTokenizer:
    SYMBOLS = ["<pad>", "<s>", "</s>", "<unk>", "|"] + list("ETAONIHSRDLUMWCFGYPBVK'XJQZ")

    def __init__(self):
        self.vocab = {s: i for i, s in enumerate(self.SYMBOLS)}
        self.blank_id = 0
        self.vocab_size = len(self.SYMBOLS)

    def decode(self, ids):
        return "".join(" " if self.SYMBOLS[i] == "|" else self.SYMBOLS[i] for i in ids if i >= 3)

    def __call__(self, text):
        return SimpleNamespace(input_ids=[self.vocab["|"] if ch == " " else self.vocab.get(ch, 3) for ch in text])


def Processor():
    from transformers import Wav2Vec2FeatureExtractor
    return SimpleNamespace(feature_extractor=Wav2Vec2FeatureExtractor())


def utterances(seed, lengths=(4000, 7000, 5200, 3100)):
    g = torch.Generator().manual_seed(seed)
    return [{'waveform': torch.randn(1, n, generator=g) * 0.1 + 0.01} for n in lengths]


def model(seed):
    from transformers import Wav2Vec2Config, Wav2Vec2ForCTC
    torch.manual_seed(seed)
    m = Wav2Vec2ForCTC(Wav2Vec2Config(**W2V2_TOY)).eval()
    with torch.no_grad():   # HF initialises biases / LayerNorm trivially and the head near zero: randomise so labels are non-empty
        for n, p in m.named_parameters():
            if p.dim() == 1 or "original0" in n:
                p.add_(0.1 * torch.randn_like(p))
        m.lm_head.weight.mul_(8.0)
        m.lm_head.bias[0] -= 0.3
    return m
