"""Tokenizers with the surface the reference loop touches: `encode(str) -> List[int]`, `decode(List[int]) -> str`,
`vocab_size()` (reference lcasr/lib.py:510,569; run_dynamic_eval_full.py:43-44).

The reference loads a SentencePiece model through the un-vendored `lcasr.utils.audio_tools.load_tokenizer()`;
`load_sentencepiece(path)` wraps any SentencePiece file the same way.  `SyntheticTokenizer` is the stand-in for
benchmarks with seeded weights (no tokenizer of the benchmark's vocabulary size exists offline): token i <-> the
word "w<i>", so decode -> encode is an exact identity and the pseudo-label text hop of the reference is preserved."""


class SyntheticTokenizer:
    def __init__(self, vocab_size):
        self._n = int(vocab_size)

    def vocab_size(self):
        return self._n

    def decode(self, ids):
        return " ".join(f"w{int(i)}" for i in ids)

    def encode(self, text):
        out = []
        for w in text.split():
            if len(w) > 1 and w[0] == "w" and w[1:].isdigit() and int(w[1:]) < self._n:
                out.append(int(w[1:]))
        return out


def load_sentencepiece(path):
    import sentencepiece as spm
    return spm.SentencePieceProcessor(model_file=path)
