"""ORACLE (test infrastructure only).  CPU restatement (plain PyTorch fp32 + autograd) of the encoder-decoder `teacher_ce`
adaptation path: `calc_loss_enc_dec` (reference lcasr/lib.py:1228-1322), `enc_dec_inference` (:1112-1134) and
`enc_dec_dynamic_eval` with `training_mode == 'teacher_ce'` (:1475-1732, update at :1638-1658), line by line where the reference
holds the code.  The MODEL is not in the reference (un-vendored `lcasr`: `get_model_class(config)`, enc_dec_dynamic_eval_test.py:45):
the encoder is oracle/conformer_ref.py, the decoder is DEFINED BY THIS BUILD and shared with the product
(dynamic-asr-eval_amd/enc_dec.py): token embedding + fixed sinusoidal positions, pre-norm blocks of causal self-attention /
cross-attention over the encoder states / SiLU FFN, final LayerNorm, linear head; bos = eos = 0 (calc_loss_enc_dec defaults
:1236-1237).  PARITY UNPINNED for the decoder architecture and for `model.generate` (greedy decode; upstream-only)."""
import math
import random

import torch
import torch.nn as nn
import torch.nn.functional as F

from .conformer_ref import SCConformerXLRef, _Lin, _Norm
from .dynamic_eval_ref import apply_masks, draw_masks, prepare_chunks

DEFAULT_DECODER = dict(dec_d_model=256, dec_layers=2, dec_heads=4, dec_ff_mult=4, dec_max_positions=2048, ctc_loss_weight=0.3)


def sinusoidal_positions(n, d):
    pos = torch.arange(n, dtype=torch.float64)[:, None]
    inv = torch.exp(torch.arange(0, d, 2, dtype=torch.float64) * (-math.log(10000.0) / d))
    tab = torch.zeros(n, d, dtype=torch.float64)
    tab[:, 0::2] = torch.sin(pos * inv)
    tab[:, 1::2] = torch.cos(pos * inv)
    return tab.float()


# ---- counter-based randomness, restated from include/dyneval.h (dyn_dropout / dyn_gumbel_argmax_rows): a draw is a pure function
# of (seed, stream, index), so the oracle reproduces the product's masks and samples exactly
import numpy as np

_M64 = (1 << 64) - 1


def mix64(seed, stream, index):
    """index: numpy uint64 array -> numpy uint64 array (splitmix64 finaliser over seed ^ stream / index multiples)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M64) ^ np.uint64(((stream + 1) * 0x9E3779B97F4A7C15) & _M64) ^ ((index + np.uint64(1)) * np.uint64(0xC2B2AE3D27D4EB4F))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def dropout_mask(shape, p, seed, stream):
    """keep-and-scale factors (0 or 1 / (1 - p) in fp32) of dyn_dropout for a contiguous tensor of `shape`."""
    n = int(np.prod(shape))
    u = (mix64(seed, stream, np.arange(n, dtype=np.uint64)) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return torch.from_numpy(np.where(u >= np.float32(p), scale, np.float32(0.0)).astype(np.float32)).reshape(shape)


def gumbel_argmax(logits, temperature, seed, step):
    """dyn_gumbel_argmax_rows for one row: argmax_c (x_c / T + g_c), g = -log(-log(u)), u = (2 (z >> 41) + 1) 2^-24."""
    C = logits.shape[-1]
    u = (np.float64(2.0) * (mix64(seed, step, np.arange(C, dtype=np.uint64)) >> np.uint64(41)).astype(np.float64) + 1.0) * 2.0 ** -24
    g = -np.log(-np.log(u))
    inv_t = np.float32(1.0) / np.float32(temperature)
    return int(np.argmax(logits.detach().numpy().astype(np.float64) * np.float64(inv_t) + g))


class _Streams:
    """The product's stream-id allocator (enc_dec.py `_next_stream`): one block of 2^20 ids per random site / sampled decode."""

    def __init__(self):
        self.draws = 0

    def next(self):
        self.draws += 1
        return self.draws << 20


class _Attn(nn.Module):
    def __init__(self, dd, d_kv, heads, cross):
        super().__init__()
        self.heads, self.cross = heads, cross
        self.norm = _Norm(dd, "layer_norm")
        if cross:
            self.q, self.kv = _Lin(dd, dd), _Lin(d_kv, 2 * dd)
        else:
            self.qkv = _Lin(dd, 3 * dd)
        self.out = _Lin(dd, dd)

    def forward(self, x, mem=None, drop=None):
        S, dd = x.shape
        H = self.heads
        n = self.norm(x)
        if self.cross:
            q = self.q(n)
            k, v = self.kv(mem).split(dd, -1)
        else:
            q, k, v = self.qkv(n).split(dd, -1)
        q, k, v = (t.view(t.shape[0], H, dd // H).transpose(0, 1) for t in (q, k, v))
        sc = q @ k.transpose(-1, -2) / math.sqrt(dd // H)
        if not self.cross:
            sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
        pm = torch.softmax(sc, -1)
        if drop is not None:                                     # (p, seed, stream): dropout on the probabilities [H, S, S]
            pm = pm * dropout_mask(pm.shape, *drop)
        o = (pm @ v).transpose(0, 1).reshape(S, dd)
        return x + self.out(o)


class _FF(nn.Module):
    def __init__(self, dd, mult):
        super().__init__()
        self.norm = _Norm(dd, "layer_norm")
        self.w1, self.w2 = _Lin(dd, dd * mult, bias=False), _Lin(dd * mult, dd, bias=False)

    def forward(self, x, drop=None):
        f = self.w2(F.silu(self.w1(self.norm(x))))
        if drop is not None:
            f = f * dropout_mask(f.shape, *drop)
        return x + f


class _DecLayer(nn.Module):
    def __init__(self, dc, d_enc):
        super().__init__()
        dd = dc["dec_d_model"]
        self.self = _Attn(dd, dd, dc["dec_heads"], False)
        self.cross = _Attn(dd, d_enc, dc["dec_heads"], True)
        self.ff = _FF(dd, dc["dec_ff_mult"])


class _Embed(nn.Module):
    def __init__(self, v, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(v, d))


class Decoder(nn.Module):
    def __init__(self, dc, d_enc, vocab):
        super().__init__()
        dd = dc["dec_d_model"]
        self.embed = _Embed(vocab, dd)
        self.layers = nn.ModuleList([_DecLayer(dc, d_enc) for _ in range(dc["dec_layers"])])
        self.norm_out = _Norm(dd, "layer_norm")
        self.head = _Lin(dd, vocab)
        self.register_buffer("pos", sinusoidal_positions(dc["dec_max_positions"], dd), persistent=False)
        # the knobs the reference's loop sets (lib.py:1519-1525,1636-1637,1703-1707); dropout only in training mode
        self.dropout_emb, self.ff_out_dropout, self.dropout_attn = 0.0, 0.0, 0.0
        self.random_seed, self.streams = 0, _Streams()

    def forward(self, tokens, mem):
        x = self.embed.weight[tokens] + self.pos[:tokens.shape[0]]
        train = self.training and torch.is_grad_enabled()
        seed = self.random_seed
        if train and self.dropout_emb > 0:                      # stream ids in the product's order: embedding, then per layer attention, FFN
            x = x * dropout_mask(x.shape, self.dropout_emb, seed, self.streams.next())
        for l in self.layers:
            d_attn = (self.dropout_attn, seed, self.streams.next()) if train and self.dropout_attn > 0 else None
            d_ff = (self.ff_out_dropout, seed, self.streams.next()) if train and self.ff_out_dropout > 0 else None
            x = l.ff(l.cross(l.self(x, drop=d_attn), mem), drop=d_ff)
        return self.head(self.norm_out(x))


class EncDecRef(nn.Module):
    def __init__(self, config=None, vocab_size=128, seed=0, blank_bias=0.0):
        super().__init__()
        config = dict(config or {})
        self.dec = {k: config.pop(k, v) for k, v in DEFAULT_DECODER.items()}
        self.encoder = SCConformerXLRef(config, vocab_size=vocab_size, seed=seed, blank_bias=blank_bias)
        self.language_model_decoder = Decoder(self.dec, self.encoder.config["d_model"], vocab_size)
        self.ctc_loss_weight = float(self.dec["ctc_loss_weight"])
        self.vocab = vocab_size
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for name, p in self.language_model_decoder.named_parameters():
                if name.endswith("norm.weight") or name.endswith("norm_out.weight"):
                    p.copy_(1.0 + 0.1 * (torch.rand(p.shape, generator=g) - 0.5))
                elif p.dim() == 1:
                    p.copy_(0.1 * (torch.rand(p.shape, generator=g) - 0.5))
                else:
                    p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) / math.sqrt(p.shape[1]) * (3.0 if name.startswith("embed") or name.startswith("head") else 1.0))

    def hip_state_dict(self):
        """Names as the product's flat buffer uses them (encoder names unprefixed, decoder under language_model_decoder.)."""
        sd = {k: v for k, v in self.encoder.state_dict().items()}
        sd.update({"language_model_decoder." + k: v for k, v in self.language_model_decoder.state_dict().items()})
        return sd

    def ordered_parameters(self):
        """Parameters in the product's flat-buffer order: encoder spec, then decoder spec."""
        return list(self.encoder.parameters()) + list(self.language_model_decoder.parameters())

    def forward(self, audio_signal, text_sequence_bos=None, a_lengths=None):
        enc = self.encoder(audio_signal)
        out = {"final_posteriors_ctc": enc["final_posteriors"], "hidden": enc["hidden"],
               "length": torch.LongTensor([enc["final_posteriors"].shape[1]] * audio_signal.shape[0])}
        if text_sequence_bos is not None:
            out["final_posteriors_lm"] = self.language_model_decoder(text_sequence_bos[0].long(), enc["hidden"][0])[None]
        return out

    @torch.no_grad()
    def generate(self, audio_signal, encoder_states=None, max_tokens=None, sample=False, temperature=1.0, seed=None):
        enc = encoder_states if encoder_states is not None else self.forward(audio_signal)
        h = enc["hidden"][0]
        limit = max_tokens if max_tokens is not None else max(1, min(self.dec["dec_max_positions"] - 1, h.shape[0] // 2))
        dec = self.language_model_decoder
        if sample:
            seed, step0 = (dec.random_seed, dec.streams.next()) if seed is None else (seed, 0)
        toks = [0]
        while len(toks) <= limit:
            logits = dec(torch.LongTensor(toks), h)
            if sample:
                nxt = gumbel_argmax(logits[-1], temperature, seed, step0 + len(toks) - 1)
            else:
                nxt = int(torch.argmax(logits[-1]).item())
            if nxt == 0:
                break
            toks.append(nxt)
        return {"text_sequence": toks[1:]}


def calc_loss_enc_dec_ref(model, audio_signal, text_sequence, a_lengths, t_lengths, bos_id=0, eos_id=0):
    """reference lcasr/lib.py:1228-1322 with its defaults (token_swap_prob 0, label_smoothing 0), batch of 1."""
    text_sequence_bos = F.pad(text_sequence, (1, 0), value=bos_id)
    target_lengths_bos = t_lengths + 1
    targets = text_sequence_bos.clone()
    targets[:, :-1] = text_sequence_bos[:, 1:].clone()
    out = model.forward(audio_signal, text_sequence_bos, a_lengths)
    ctc_out, lm_out, a_length_out = out['final_posteriors_ctc'], out['final_posteriors_lm'], out['length']
    if model.ctc_loss_weight > 0.0:
        ctc_o = ctc_out.repeat(t_lengths.shape[0], 1, 1).transpose(0, 1)            # 'b n c -> n b c'
        ctc_loss = F.ctc_loss(log_probs=ctc_o, targets=text_sequence, input_lengths=a_length_out.repeat(t_lengths.shape[0]),
                              target_lengths=t_lengths, reduction='sum', blank=ctc_o.shape[-1] - 1)
        ctc_loss_to_bwd = ctc_loss / (ctc_o.shape[1] * ctc_o.shape[0]) * 100
    else:
        ctc_loss_to_bwd = 0
    assert target_lengths_bos.max() == target_lengths_bos.min()
    targets[:, -1] = 0
    lm_loss = F.cross_entropy(input=lm_out.reshape(-1, lm_out.shape[-1]), target=targets.reshape(-1), ignore_index=-100, reduction='sum')
    lm_loss_to_bwd = lm_loss / (lm_out.shape[0] * lm_out.shape[1])
    return ctc_loss_to_bwd * model.ctc_loss_weight + lm_loss_to_bwd * (1 - model.ctc_loss_weight)


def enc_dec_inference_ref(model, spec, seq_len, overlap, tokenizer):
    """reference lcasr/lib.py:1112-1134"""
    assert overlap == 0
    training_data, training_keys = prepare_chunks(spec, seq_len, overlap)
    texts = []
    for k in training_keys:
        toks = model.generate(training_data[k])["text_sequence"]
        texts.append(tokenizer.decode(toks).strip())
    return " ".join(texts).replace('  ', ' ').strip()


def enc_dec_dynamic_eval_ref(model, spec, seq_len, tokenizer, optimizer_cls, lr_args, epochs=1, fixed_masks=None, return_params=False,
                             skip_fn=None, dropout_emb=0.0, dropout_post_ff=0.0, dropout_attn=0.0, agreement_temperature=None,
                             random_seed=0, trace=None):
    """reference lcasr/lib.py:1475-1732, training_mode 'teacher_ce', filters applied through `skip_fn(tokens, text, frames[,
    agreement_text])`; `agreement_temperature` switches the sampled second decode of the decode-agreement filter on (:1620-1627);
    the three dropout knobs as the reference sets them (:1511-1525,1636-1637,1703-1707).  `trace` collects (teacher, agreement) texts."""
    spec_n = spec.shape[-1]
    dec = model.language_model_decoder
    dec.dropout_emb, dec.ff_out_dropout, dec.dropout_attn = dropout_emb, dropout_post_ff, 0.0
    dec.random_seed, dec.streams = random_seed, _Streams()
    original = [p.clone().detach() for p in model.ordered_parameters()]
    optimizer = optimizer_cls(model.ordered_parameters(), **lr_args)
    overlap = 0
    if seq_len > spec_n:
        seq_len = spec_n
    model.eval()
    training_data, training_keys = prepare_chunks(spec, seq_len, overlap)
    for epoch in range(epochs):
        for idx in range(len(training_keys)):
            key = training_keys[idx]
            audio_chunk = training_data[key].clone().repeat(2, 1, 1)
            masks = fixed_masks[key] if fixed_masks is not None else (draw_masks(0, 1, 80), ([], []))
            apply_masks(audio_chunk[0], masks, False)
            with torch.no_grad():
                enc_states = model.forward(audio_signal=audio_chunk[-1, None])
            teacher_tokens = model.generate(audio_chunk[-1, None], encoder_states=enc_states)["text_sequence"]
            teacher_pred = torch.tensor(teacher_tokens, dtype=torch.long)
            teacher_text = tokenizer.decode(teacher_tokens).strip()
            agreement_text = None
            if agreement_temperature is not None:
                agreement_text = tokenizer.decode(model.generate(audio_chunk[-1, None], encoder_states=enc_states, sample=True,
                                                                 temperature=agreement_temperature)["text_sequence"]).strip()
            if trace is not None:
                trace.append((teacher_text, agreement_text))
            if skip_fn is not None:
                skip = skip_fn(teacher_tokens, teacher_text, audio_chunk.shape[-1], agreement_text) if agreement_temperature is not None \
                    else skip_fn(teacher_tokens, teacher_text, audio_chunk.shape[-1])
                if skip:
                    continue
            dec.dropout_attn = dropout_attn
            dec.train()
            loss = calc_loss_enc_dec_ref(model, audio_chunk[:1], teacher_pred[None, :], torch.LongTensor([audio_chunk.shape[-1]]),
                                         torch.LongTensor([teacher_pred.shape[-1]]))
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            dec.eval()
            dec.dropout_attn = 0.0
    model.eval()
    final_out = enc_dec_inference_ref(model, spec, seq_len, overlap, tokenizer)
    updated = [p.clone().detach() for p in model.ordered_parameters()] if return_params else None
    with torch.no_grad():
        for p, po in zip(model.ordered_parameters(), original):
            p.copy_(po)
    return (final_out, updated) if return_params else final_out
