"""Encoder-decoder test-time adaptation, `teacher_ce` mode: mirror of the reference's
`calc_loss_enc_dec` (lcasr/lib.py:1228-1322), `enc_dec_inference` (:1112-1134), `generate_enc_dec` call sites (:1128) and
`enc_dec_dynamic_eval` (:1475-1732), same names / argument meaning / printed lines, on the HIP kernels.

The reference takes the model from the un-vendored `lcasr` package (`get_model_class(config)`, an SCConformerXL encoder with a
CTC head plus `language_model_decoder`; enc_dec_dynamic_eval_test.py:45-46).  Only its call surface is visible in the reference:
`model.forward(audio_signal[, text_sequence_bos, a_lengths]) -> {'final_posteriors_ctc', 'final_posteriors_lm', 'length'}`,
`model.generate(audio, encoder_states=...)["text_sequence"]`, `model.ctc_loss_weight`, `model.language_model_decoder.{pos_enc,
layers, dropout_emb, ff_out_dropout}`, `model.pos_enc`, `model.ctc_decoder.num_classes`.  The DECODER ARCHITECTURE IS DEFINED BY THIS
BUILD (parity unpinned, like the encoder's internals): token embedding + fixed sinusoidal positions, `dec_layers` pre-norm blocks of
causal self-attention / cross-attention over the encoder states / SiLU FFN, final LayerNorm, linear head over the tokenizer
vocabulary; bos = eos = 0 as in calc_loss_enc_dec's defaults (:1236-1237).  oracle/enc_dec_ref.py holds the same definition in torch.

Scope: `training_mode == 'teacher_ce'` (:1638-1658) with every flag of that path: the teacher filters incl. the sampled-decode
agreement filter (`model.generate(sample=True, temperature=...)`, :1620-1627) and the decoder dropout knobs `dropout_emb /
dropout_post_ff / dropout_attn` (:1511-1522,1636-1637,1703-1707).  Randomness is counter-based (dyn_dropout / dyn_gumbel_argmax_rows:
a draw is a pure function of (seed, stream, index)), so it is reproducible and the oracle restates it exactly.  The RL modes
(`grpo`, `maxrl`: sampled rollouts, reward models) are out of scope and raise.  Decoder parameters live in the encoder's flat
buffers (SCConformerXL(extra_spec=...)), so snapshot / restore / MADGRAD step are the same single operations as on the CTC path."""
import ctypes
import math
import os
import random
from types import SimpleNamespace

import torch

from . import ops
from ._lib import DEC_PTRS_PER_LAYER, DecoderDesc, check, load
from .augment import SpecAugment
from .decoding import GreedyCTCDecoder
from .enc_dec_teacher_filters import should_skip_faulty_teacher_prediction
from .lib import _window_fill_value, get_lr_args_from_args, get_specaugment_config_from_args, prepare_chunks
from .model import SCConformerXL
from .optim import MADGRAD

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, **_):
        return x

DEC = "language_model_decoder."
DEFAULT_DECODER = dict(dec_d_model=256, dec_layers=2, dec_heads=4, dec_ff_mult=4, dec_max_positions=2048, ctc_loss_weight=0.3)


def decoder_spec(dc, d_enc, vocab):
    dd, ff = dc["dec_d_model"], dc["dec_d_model"] * dc["dec_ff_mult"]
    spec = [(DEC + "embed.weight", (vocab, dd))]
    for l in range(dc["dec_layers"]):
        p = f"{DEC}layers.{l}."
        spec += [(p + "self.norm.weight", (dd,)), (p + "self.norm.bias", (dd,)), (p + "self.qkv.weight", (3 * dd, dd)),
                 (p + "self.qkv.bias", (3 * dd,)), (p + "self.out.weight", (dd, dd)), (p + "self.out.bias", (dd,)),
                 (p + "cross.norm.weight", (dd,)), (p + "cross.norm.bias", (dd,)), (p + "cross.q.weight", (dd, dd)),
                 (p + "cross.q.bias", (dd,)), (p + "cross.kv.weight", (2 * dd, d_enc)), (p + "cross.kv.bias", (2 * dd,)),
                 (p + "cross.out.weight", (dd, dd)), (p + "cross.out.bias", (dd,)),
                 (p + "ff.norm.weight", (dd,)), (p + "ff.norm.bias", (dd,)), (p + "ff.w1.weight", (ff, dd)), (p + "ff.w2.weight", (dd, ff))]
    spec += [(DEC + "norm_out.weight", (dd,)), (DEC + "norm_out.bias", (dd,)), (DEC + "head.weight", (vocab, dd)), (DEC + "head.bias", (vocab,))]
    return spec


def sinusoidal_positions(n, d):
    """Fixed positional table [n, d] (float64 -> fp32 once; shared with the oracle).  Frozen: the reference freezes
    `language_model_decoder.pos_enc` (lib.py:1505-1509,1533-1536) — here it is not a parameter at all."""
    pos = torch.arange(n, dtype=torch.float64)[:, None]
    inv = torch.exp(torch.arange(0, d, 2, dtype=torch.float64) * (-math.log(10000.0) / d))
    tab = torch.zeros(n, d, dtype=torch.float64)
    tab[:, 0::2] = torch.sin(pos * inv)
    tab[:, 1::2] = torch.cos(pos * inv)
    return tab.float()


class _NoParams:
    def parameters(self):
        return []


class _AttnFn:
    """`layer[0].fn.dropout_p` of the reference's decoder layers (lib.py:1525,1636,1706): the self-attention dropout rate."""

    def __init__(self):
        self.dropout_p = 0.0


class _DecoderKnobs:
    """What the reference touches on `model.language_model_decoder`: `.pos_enc`, `.layers[i][0].fn.dropout_p`, `.dropout_emb`,
    `.ff_out_dropout`, `.train()` / `.eval()` (dropout is applied in training mode only, as nn.Dropout)."""

    def __init__(self, n_layers):
        self.pos_enc = _NoParams()
        self.layers = [[SimpleNamespace(fn=_AttnFn())] for _ in range(n_layers)]
        self.dropout_emb = 0.0
        self.ff_out_dropout = 0.0
        self.training = False

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)


def _check_ids(ids, vocab, what):
    """Token ids are validated on the host before they reach the embedding / loss kernels (an id outside the decoder
    vocabulary, e.g. a tokenizer / vocab_size mismatch, must fail loudly instead of reading out of bounds)."""
    ids = [int(i) for i in ids]
    if ids and (min(ids) < 0 or max(ids) >= vocab):
        raise ops.DynError(f"{what}: token id outside [0, {vocab}) (min {min(ids)}, max {max(ids)}): tokenizer / vocab_size mismatch?")
    return ids


class EncDecSCConformerXL(SCConformerXL):
    """Encoder (this package's SCConformerXL, CTC head = `ctc_decoder`) + autoregressive decoder."""

    def __init__(self, config=None, vocab_size=128, device="cuda:0"):
        config = dict(config or {})
        self.dec = {k: config.pop(k, v) for k, v in DEFAULT_DECODER.items()}
        d_enc = config.get("d_model", 768)
        if self.dec["dec_d_model"] % 256 or self.dec["dec_d_model"] % self.dec["dec_heads"]:
            raise ValueError("dec_d_model must be a multiple of 256 (wave-per-row norm kernels) and of dec_heads")
        super().__init__(config, vocab_size=vocab_size, device=device, extra_spec=decoder_spec(self.dec, d_enc, vocab_size))
        self.vocab = vocab_size
        self.ctc_loss_weight = float(self.dec["ctc_loss_weight"])
        self.ctc_decoder = self.decoder                          # the reference's name for the CTC head (lib.py:1559)
        self.pos_table = sinusoidal_positions(self.dec["dec_max_positions"], self.dec["dec_d_model"]).to(self.device)
        # the attributes the reference's loop sets on `model.language_model_decoder` (lib.py:1519-1522,1636-1637,1703-1707)
        self.language_model_decoder = _DecoderKnobs(self.dec["dec_layers"])
        self.pos_enc = _NoParams()
        self.use_graphs = False
        self._dctx = None
        # generate(): the lean one-token kernels of dyn_decoder_steps (default) or, with DYN_FUSED_DECODE=0, the tile kernels at M = 1
        hd = self.dec["dec_d_model"] // self.dec["dec_heads"]
        self.fused_decode = (os.environ.get("DYN_FUSED_DECODE", "1") != "0" and self.dec["dec_d_model"] * self.dec["dec_ff_mult"] <= 2048
                             and 4 <= hd <= 256 and hd & (hd - 1) == 0)
        self.random_seed = 0          # seed of the counter-based dropout / sampling draws (args.random_seed in the loop)
        self._draws = 0               # one stream id per random site and call: no draw is ever reused

    # ------------------------------------------------------------------ reference call surface
    def forward(self, audio_signal, text_sequence_bos=None, a_lengths=None):
        with ops.use_workspace(self._scratch()):
            enc = self._forward_eager(audio_signal)
            out = {"final_posteriors_ctc": enc["final_posteriors"], "hidden": self._hidden,
                   "length": torch.full((audio_signal.shape[0],), enc["final_posteriors"].shape[1], dtype=torch.int32, device=self.device)}
            if text_sequence_bos is not None:
                if audio_signal.shape[0] != 1 or text_sequence_bos.shape[0] != 1:
                    raise ops.DynError("enc-dec forward with text: batch size 1 only (num_negatives == 1 in the reference, lib.py:1494)")
                out["final_posteriors_lm"] = self._decoder_forward(text_sequence_bos[0].to(torch.int32), self._hidden[0])[None]
        return out

    __call__ = forward

    def generate(self, audio_signal, encoder_states=None, sample=False, temperature=1.0, max_tokens=None, seed=None, check_every=8):
        """Autoregressive decode of ONE window -> {'text_sequence': [ids]} (reference call sites lib.py:1128,1579-1582,1620-1625):
        greedy, or with `sample=True` a draw from softmax(logits / temperature) per step (Gumbel-max on counter-based uniforms:
        dyn_gumbel_argmax_rows; `seed` defaults to a fresh stream of this model's random_seed).
        Incremental: every step runs ONE token through the decoder against per-layer caches — the self-attention keys / values of
        the prefix live in a packed [limit + 1, 3 * dd] buffer the QKV projection writes row t of, the cross-attention keys /
        values of the encoder states are projected once.  The chosen id goes straight into the device token buffer; the host
        looks for eos only every `check_every` steps (ids after the first eos are never read), so there is no per-token sync."""
        with torch.no_grad():
            enc = encoder_states if encoder_states is not None else self.forward(audio_signal)
            h = enc["hidden"][0]
            limit = max_tokens if max_tokens is not None else max(1, min(self.dec["dec_max_positions"] - 1, h.shape[0] // 2))
            dc, P = self.dec, self.P
            dd, L = dc["dec_d_model"], dc["dec_layers"]
            if sample:
                if not temperature > 0.0:
                    raise ops.DynError("generate(sample=True): temperature must be > 0")
                if seed is None:
                    seed, step0 = self.random_seed, self._next_stream()
                else:
                    step0 = 0
            st = torch.cuda.current_stream().cuda_stream
            with ops.use_workspace(self._scratch()):
                kv = [ops.linear(h, P[f"{DEC}layers.{l}.cross.kv.weight"], P[f"{DEC}layers.{l}.cross.kv.bias"]) for l in range(L)]
                cache = [torch.empty(limit + 1, 3 * dd, device=self.device, dtype=torch.float32) for _ in range(L)]
                tok_dev = torch.zeros(limit + 2, dtype=torch.int32, device=self.device)           # tok_dev[0] = bos
                n_tok, t = None, 0
                fused = self._decoder_desc(kv, cache, tok_dev, h.shape[0]) if self.fused_decode else None
                while t < limit and n_tok is None:
                    n = min(check_every, limit - t)
                    if fused is not None:                                                          # one C call, 8 * layers + 2 launches per token
                        check(load().dyn_decoder_steps(ctypes.byref(fused[0]), t, n, 1 if sample else 0, 1.0 / float(temperature) if sample else 1.0,
                                                       int(seed) if sample else 0, int(step0) if sample else 0, st), "dyn_decoder_steps")
                        t += n
                    for _ in range(0 if fused is not None else n):                                 # the tile-kernel path (DYN_FUSED_DECODE=0)
                        logits = self._decoder_step(tok_dev, t, h, kv, cache)                     # [1, V]
                        nxt = tok_dev[t + 1:]
                        if sample:
                            check(load().dyn_gumbel_argmax_rows(logits.data_ptr(), 1, self.vocab, self.vocab, 1.0 / float(temperature), int(seed),
                                                                int(step0 + t), nxt.data_ptr(), st), "dyn_gumbel_argmax_rows")
                        else:
                            check(load().dyn_argmax_rows(logits.data_ptr(), 1, self.vocab, self.vocab, nxt.data_ptr(), 0, st), "dyn_argmax_rows")
                        t += 1
                    got = tok_dev[1:t + 1].tolist()                                                # one sync per `check_every` tokens
                    if 0 in got:                                                                   # eos
                        n_tok = got.index(0)
                toks = tok_dev[1:t + 1].tolist()[:n_tok] if n_tok is not None else tok_dev[1:t + 1].tolist()
        return {"text_sequence": toks}

    def _decoder_desc(self, kv, cache, tok_dev, n_enc):
        """dyn_decoder_desc of one generate() call (include/dyneval.h): returns (descriptor, objects it points into)."""
        P, dc = self.P, self.dec
        dd, L = dc["dec_d_model"], dc["dec_layers"]
        ff = dd * dc["dec_ff_mult"]
        names = ("self.norm.weight", "self.norm.bias", "self.qkv.weight", "self.qkv.bias", "self.out.weight", "self.out.bias",
                 "cross.norm.weight", "cross.norm.bias", "cross.q.weight", "cross.q.bias", "cross.out.weight", "cross.out.bias",
                 "ff.norm.weight", "ff.norm.bias", "ff.w1.weight", "ff.w2.weight")
        ptrs = (ctypes.c_void_p * (L * DEC_PTRS_PER_LAYER))()
        for l in range(L):
            for i, nm in enumerate(names):
                ptrs[l * DEC_PTRS_PER_LAYER + i] = P[f"{DEC}layers.{l}.{nm}"].data_ptr()
            ptrs[l * DEC_PTRS_PER_LAYER + 16] = cache[l].data_ptr()
            ptrs[l * DEC_PTRS_PER_LAYER + 17] = kv[l].data_ptr()
        logits = torch.empty(self.vocab, device=self.device, dtype=torch.float32)
        scratch = torch.empty(6 * dd + ff + 16 * dc["dec_heads"], device=self.device, dtype=torch.float32)
        d = DecoderDesc(d_model=dd, heads=dc["dec_heads"], d_ff=ff, vocab=self.vocab, layers=L, n_enc=int(n_enc),
                        max_positions=dc["dec_max_positions"], eps=float(self.config["norm_eps"]),
                        embed=P[DEC + "embed.weight"].data_ptr(), pos_table=self.pos_table.data_ptr(),
                        norm_out_w=P[DEC + "norm_out.weight"].data_ptr(), norm_out_b=P[DEC + "norm_out.bias"].data_ptr(),
                        head_w=P[DEC + "head.weight"].data_ptr(), head_b=P[DEC + "head.bias"].data_ptr(),
                        layer_ptrs=ctypes.cast(ptrs, ctypes.POINTER(ctypes.c_void_p)), tokens=tok_dev.data_ptr(), logits=logits.data_ptr(),
                        scratch=scratch.data_ptr(), scratch_floats=scratch.numel())
        return d, (ptrs, logits, scratch)

    def _next_stream(self):
        """A fresh block of 2^20 stream ids for one random site (dropout mask) or one sampled decode (one id per step)."""
        self._draws += 1
        return self._draws << 20

    def _dropout(self, x, p, stream_id, out=None):
        """y = dropout(x) with the mask of (random_seed, stream_id); the backward is the same call on the gradient."""
        out = torch.empty_like(x) if out is None else out
        check(load().dyn_dropout(x.data_ptr(), out.data_ptr(), x.numel(), float(p), int(self.random_seed), int(stream_id),
                                 torch.cuda.current_stream().cuda_stream), "dyn_dropout")
        return out

    def _decoder_step(self, tok_dev, t, h_enc, kv, cache):
        """One token (position t, id tok_dev[t]) through the decoder with cached keys / values -> logits [1, V]."""
        P, dc = self.P, self.dec
        dd, eps = dc["dec_d_model"], self.config["norm_eps"]
        Tk = h_enc.shape[0]
        st = torch.cuda.current_stream().cuda_stream
        x = torch.empty(1, dd, device=self.device, dtype=torch.float32)
        check(load().dyn_embedding_fwd(tok_dev[t:].data_ptr(), P[DEC + "embed.weight"].data_ptr(), self.pos_table[t:].data_ptr(), x.data_ptr(), 1, dd,
                                       self.vocab, 1, st), "dyn_embedding_fwd")
        for l in range(dc["dec_layers"]):
            p = f"{DEC}layers.{l}."
            c = cache[l]
            n1, _, _ = ops.layernorm(x, P[p + "self.norm.weight"], P[p + "self.norm.bias"], eps)
            row = ops.linear(n1, P[p + "self.qkv.weight"], P[p + "self.qkv.bias"], out=c[t:t + 1])          # q | k | v of position t
            o1, _ = self._attend(row, c[:, dd:], c[:, 2 * dd:], 1, t + 1, 3 * dd, 3 * dd, False, False)        # the prefix IS the causal mask
            x1 = ops.linear(o1, P[p + "self.out.weight"], P[p + "self.out.bias"], beta=1.0, residual=x)
            n2, _, _ = ops.layernorm(x1, P[p + "cross.norm.weight"], P[p + "cross.norm.bias"], eps)
            q2 = ops.linear(n2, P[p + "cross.q.weight"], P[p + "cross.q.bias"])
            o2, _ = self._attend(q2, kv[l], kv[l][:, dd:], 1, Tk, dd, 2 * dd, False, False)
            x2 = ops.linear(o2, P[p + "cross.out.weight"], P[p + "cross.out.bias"], beta=1.0, residual=x1)
            n3, _, _ = ops.layernorm(x2, P[p + "ff.norm.weight"], P[p + "ff.norm.bias"], eps)
            a = ops.silu(ops.linear(n3, P[p + "ff.w1.weight"]))
            x = ops.linear(a, P[p + "ff.w2.weight"], beta=1.0, residual=x2)
        nf, _, _ = ops.layernorm(x, P[DEC + "norm_out.weight"], P[DEC + "norm_out.bias"], eps)
        return ops.linear(nf, P[DEC + "head.weight"], P[DEC + "head.bias"])

    # ------------------------------------------------------------------ decoder forward / backward (B = 1)
    def _attend(self, q, k, v, S, Tk, ldq, ldk, causal, save, drop=None):
        """softmax(q k^T / sqrt(hd)) v per head; q rows have leading dimension ldq, k / v rows ldk (views into packed projections).
        `drop` = (p, stream id): dropout on the probabilities (training mode); returns (O, P[, dropped P])."""
        Hh, dd = self.dec["dec_heads"], self.dec["dec_d_model"]
        hd = dd // Hh
        Pm = torch.empty(Hh, S, Tk, device=self.device, dtype=torch.float32)
        ops.gemm(q, k, Pm, trans_b=True, M=S, N=Tk, K=hd, lda=ldq, ldb=ldk, ldc=Tk, nb1=1, nb2=Hh, sa=(0, hd), sb=(0, hd), sc=(0, S * Tk),
                 alpha=1.0 / math.sqrt(hd))
        if causal:
            check(load().dyn_causal_mask(Pm.data_ptr(), Hh, S, torch.cuda.current_stream().cuda_stream), "dyn_causal_mask")
        ops.softmax(Pm, out=Pm)
        Pd = self._dropout(Pm, drop[0], drop[1]) if drop is not None else Pm
        O = torch.empty(S, dd, device=self.device, dtype=torch.float32)
        ops.gemm(Pd, v, O, M=S, N=hd, K=Tk, lda=Tk, ldb=ldk, ldc=dd, nb1=1, nb2=Hh, sa=(0, S * Tk), sb=(0, hd), sc=(0, hd))
        if drop is not None:
            return O, ((Pm, Pd) if save else None)
        return O, (Pm if save else None)

    def _attend_bwd(self, dO, Pm, q, k, v, dq, dk, dv, S, Tk, ldq, ldk, drop=None):
        """Gradients of _attend into the (strided) dq / dk / dv views of the packed projection gradients; each view is written
        exactly once (beta = 0).  With dropout `Pm` = (probabilities, dropped probabilities) and `drop` = (p, stream id)."""
        Hh, dd = self.dec["dec_heads"], self.dec["dec_d_model"]
        hd = dd // Hh
        sP = (0, S * Tk)
        Pm, Pd = Pm if drop is not None else (Pm, Pm)
        ops.gemm(Pd, dO, dv, trans_a=True, M=Tk, N=hd, K=S, lda=Tk, ldb=dd, ldc=ldk, nb1=1, nb2=Hh, sa=sP, sb=(0, hd), sc=(0, hd))
        dP = torch.empty_like(Pm)
        ops.gemm(dO, v, dP, trans_b=True, M=S, N=Tk, K=hd, lda=dd, ldb=ldk, ldc=Tk, nb1=1, nb2=Hh, sa=(0, hd), sb=(0, hd), sc=sP)
        if drop is not None:
            self._dropout(dP, drop[0], drop[1], out=dP)            # same mask, same 1 / (1 - p)
        ops.softmax_bwd(Pm, dP, out=dP, scale=1.0)
        sc = 1.0 / math.sqrt(hd)
        ops.gemm(dP, k, dq, M=S, N=hd, K=Tk, lda=Tk, ldb=ldk, ldc=ldq, nb1=1, nb2=Hh, sa=sP, sb=(0, hd), sc=(0, hd), alpha=sc)
        ops.gemm(dP, q, dk, trans_a=True, M=Tk, N=hd, K=S, lda=Tk, ldb=ldq, ldc=ldk, nb1=1, nb2=Hh, sa=sP, sb=(0, hd), sc=(0, hd), alpha=sc)

    def _decoder_forward(self, tokens, h_enc, cached_kv=None):
        """tokens int32 [S] (bos first), h_enc [T', d_enc] -> logits [S, V].  Saves activations in grad mode."""
        P, dc = self.P, self.dec
        dd, eps = dc["dec_d_model"], self.config["norm_eps"]
        S, Tk = tokens.shape[0], h_enc.shape[0]
        if S > dc["dec_max_positions"]:
            raise ops.DynError(f"decoder sequence of {S} tokens exceeds dec_max_positions {dc['dec_max_positions']}")
        save = torch.is_grad_enabled() and cached_kv is None
        st = torch.cuda.current_stream().cuda_stream
        x = torch.empty(S, dd, device=self.device, dtype=torch.float32)
        check(load().dyn_embedding_fwd(tokens.data_ptr(), P[DEC + "embed.weight"].data_ptr(), self.pos_table.data_ptr(), x.data_ptr(), S, dd,
                                       self.vocab, self.pos_table.shape[0], st), "dyn_embedding_fwd")
        ctx = {"tokens": tokens, "h_enc": h_enc, "layers": []} if save else None
        # dropout: only in the decoder's training mode (the reference brackets the supervised step with .train() / .eval(),
        # lib.py:1637,1703) and only for a forward that will be differentiated
        knobs = self.language_model_decoder
        train = save and knobs.training
        p_emb = float(knobs.dropout_emb) if train else 0.0
        p_ff = float(knobs.ff_out_dropout) if train else 0.0
        d_emb = (p_emb, self._next_stream()) if p_emb > 0.0 else None
        if d_emb is not None:
            self._dropout(x, *d_emb, out=x)
        if save:
            ctx["d_emb"] = d_emb
        for l in range(dc["dec_layers"]):
            p = f"{DEC}layers.{l}."
            lc = {}
            p_attn = float(knobs.layers[l][0].fn.dropout_p) if train else 0.0
            d_attn = (p_attn, self._next_stream()) if p_attn > 0.0 else None
            d_ff = (p_ff, self._next_stream()) if p_ff > 0.0 else None
            n1, m1, r1 = ops.layernorm(x, P[p + "self.norm.weight"], P[p + "self.norm.bias"], eps)
            qkv = ops.linear(n1, P[p + "self.qkv.weight"], P[p + "self.qkv.bias"])
            o1, P1 = self._attend(qkv, qkv[:, dd:], qkv[:, 2 * dd:], S, S, 3 * dd, 3 * dd, True, save, drop=d_attn)
            x1 = ops.linear(o1, P[p + "self.out.weight"], P[p + "self.out.bias"], beta=1.0, residual=x)
            n2, m2, r2 = ops.layernorm(x1, P[p + "cross.norm.weight"], P[p + "cross.norm.bias"], eps)
            q2 = ops.linear(n2, P[p + "cross.q.weight"], P[p + "cross.q.bias"])
            kv = cached_kv[l] if cached_kv is not None else ops.linear(h_enc, P[p + "cross.kv.weight"], P[p + "cross.kv.bias"])
            o2, P2 = self._attend(q2, kv, kv[:, dd:], S, Tk, dd, 2 * dd, False, save)
            x2 = ops.linear(o2, P[p + "cross.out.weight"], P[p + "cross.out.bias"], beta=1.0, residual=x1)
            n3, m3, r3 = ops.layernorm(x2, P[p + "ff.norm.weight"], P[p + "ff.norm.bias"], eps)
            u = ops.linear(n3, P[p + "ff.w1.weight"])
            a = ops.silu(u)
            if d_ff is None:
                x3 = ops.linear(a, P[p + "ff.w2.weight"], beta=1.0, residual=x2)
            else:                                   # x3 = x2 + dropout(a W2^T)
                x3 = self._dropout(ops.linear(a, P[p + "ff.w2.weight"]), *d_ff)
                ops.axpby(x2, x3, a=1.0, b=1.0)
            if save:
                lc = dict(x=x, n1=n1, m1=m1, r1=r1, qkv=qkv, P1=P1, o1=o1, x1=x1, n2=n2, m2=m2, r2=r2, q2=q2, kv=kv, P2=P2, o2=o2, x2=x2,
                          n3=n3, m3=m3, r3=r3, u=u, a=a, d_attn=d_attn, d_ff=d_ff)
                ctx["layers"].append(lc)
            x = x3
        nf, mf, rf = ops.layernorm(x, P[DEC + "norm_out.weight"], P[DEC + "norm_out.bias"], eps)
        logits = ops.linear(nf, P[DEC + "head.weight"], P[DEC + "head.bias"])
        if save:
            ctx["final"] = (x, nf, mf, rf)
            self._dctx = ctx
        return logits

    def _wb(self, dy, x, name, bias=True):
        """weight (+ bias) gradient of y = x W^T + b, accumulated into the flat gradient buffer."""
        if self.trainable(name + ".weight"):
            ops.linear_wgrad(dy, x, self.G[name + ".weight"], beta=1.0)
        if bias and self.trainable(name + ".bias"):
            ops.colsum(dy, self.G[name + ".bias"], beta=1.0)

    def _decoder_backward(self, dlogits):
        """dL/dlogits [S, V] -> accumulates every decoder parameter gradient; returns dL/dh_enc [T', d_enc]."""
        P, G, dc, ctx = self.P, self.G, self.dec, self._dctx
        dd = dc["dec_d_model"]
        tokens, h_enc = ctx["tokens"], ctx["h_enc"]
        S, Tk = tokens.shape[0], h_enc.shape[0]
        x, nf, mf, rf = ctx["final"]
        self._wb(dlogits, nf, DEC + "head")
        dn = ops.linear_dgrad(dlogits, P[DEC + "head.weight"])
        dx = torch.empty_like(x)
        ops.layernorm_bwd(x, P[DEC + "norm_out.weight"], mf, rf, dn, dx, G[DEC + "norm_out.weight"], G[DEC + "norm_out.bias"], dx_beta=0.0)
        dh_enc = torch.zeros_like(h_enc)
        for l in reversed(range(dc["dec_layers"])):
            p = f"{DEC}layers.{l}."
            c = ctx["layers"][l]
            # FFN
            df = dx if c["d_ff"] is None else self._dropout(dx, *c["d_ff"])          # gradient of the (dropped) FFN branch; dx stays the residual's
            if self.trainable(p + "ff.w2.weight"):
                ops.linear_wgrad(df, c["a"], G[p + "ff.w2.weight"], beta=1.0)
            du = ops.silu_bwd(c["u"], ops.linear_dgrad(df, P[p + "ff.w2.weight"]))
            if self.trainable(p + "ff.w1.weight"):
                ops.linear_wgrad(du, c["n3"], G[p + "ff.w1.weight"], beta=1.0)
            dn3 = ops.linear_dgrad(du, P[p + "ff.w1.weight"])
            ops.layernorm_bwd(c["x2"], P[p + "ff.norm.weight"], c["m3"], c["r3"], dn3, dx, G[p + "ff.norm.weight"], G[p + "ff.norm.bias"], dx_beta=1.0)
            # cross-attention
            self._wb(dx, c["o2"], p + "cross.out")
            do2 = ops.linear_dgrad(dx, P[p + "cross.out.weight"])
            dq2 = torch.empty_like(c["q2"])
            dkv = torch.empty_like(c["kv"])
            self._attend_bwd(do2, c["P2"], c["q2"], c["kv"], c["kv"][:, dd:], dq2, dkv, dkv[:, dd:], S, Tk, dd, 2 * dd)
            self._wb(dkv, h_enc, p + "cross.kv")
            ops.linear_dgrad(dkv, P[p + "cross.kv.weight"], out=dh_enc, beta=1.0)
            self._wb(dq2, c["n2"], p + "cross.q")
            dn2 = ops.linear_dgrad(dq2, P[p + "cross.q.weight"])
            ops.layernorm_bwd(c["x1"], P[p + "cross.norm.weight"], c["m2"], c["r2"], dn2, dx, G[p + "cross.norm.weight"], G[p + "cross.norm.bias"],
                              dx_beta=1.0)
            # causal self-attention
            self._wb(dx, c["o1"], p + "self.out")
            do1 = ops.linear_dgrad(dx, P[p + "self.out.weight"])
            dqkv = torch.empty_like(c["qkv"])
            qkv = c["qkv"]
            self._attend_bwd(do1, c["P1"], qkv, qkv[:, dd:], qkv[:, 2 * dd:], dqkv, dqkv[:, dd:], dqkv[:, 2 * dd:], S, S, 3 * dd, 3 * dd,
                             drop=c["d_attn"])
            self._wb(dqkv, c["n1"], p + "self.qkv")
            dn1 = ops.linear_dgrad(dqkv, P[p + "self.qkv.weight"])
            ops.layernorm_bwd(c["x"], P[p + "self.norm.weight"], c["m1"], c["r1"], dn1, dx, G[p + "self.norm.weight"], G[p + "self.norm.bias"],
                              dx_beta=1.0)
        if ctx["d_emb"] is not None:
            self._dropout(dx, *ctx["d_emb"], out=dx)
        if self.trainable(DEC + "embed.weight"):
            check(load().dyn_embedding_bwd(tokens.data_ptr(), dx.data_ptr(), G[DEC + "embed.weight"].data_ptr(), S, dd, self.vocab, 1.0,
                                           torch.cuda.current_stream().cuda_stream), "dyn_embedding_bwd")
        self._dctx = None
        return dh_enc


# ------------------------------------------------------------------------------------------------ the reference's functions
def calc_loss_enc_dec(model, audio_signal, text_sequence, a_lengths, t_lengths, tokenizer, token_swap_prob=0.0, bos_id=0, eos_id=0,
                      label_smoothing=0.0, backward=True):
    """reference lcasr/lib.py:1228-1322 for one sample (num_negatives == 1): bos-prefixed inputs, targets shifted left with
    eos (= 0) in the last place (:1240-1244,1277), `model.forward(audio, text_bos, a_lengths)` (:1255), CTC term
    F.ctc_loss(sum, blank = C - 1) / (N * B) * 100 (:1258-1272), LM term F.cross_entropy(sum) / (B * S) (:1290-1299), mixed with
    `model.ctc_loss_weight` (:1303).  With `backward` the gradients of that loss are accumulated into the model's flat gradient
    buffer (decoder backward -> encoder backward with the cross-attention's gradient on the encoder states): the reference's
    `loss.backward()` at :1657.  Returns the same dict (loss as a python float, no autograd graph exists here)."""
    if token_swap_prob > 0.0 or label_smoothing > 0.0:
        raise NotImplementedError("token_swap_prob / label_smoothing: defaults only (the teacher_ce path passes neither, lib.py:1645-1652)")
    dev = model.device
    text = text_sequence.to(dev)
    assert text.shape[0] == 1 and audio_signal.shape[0] == 1, "calc_loss_enc_dec: batch of 1 (num_negatives == 1)"
    S = int(t_lengths[0])
    _check_ids(text_sequence[0, :S].tolist(), model.vocab, "calc_loss_enc_dec: text_sequence")
    text_bos = torch.zeros(1, S + 1, dtype=torch.int32, device=dev)
    text_bos[:, 0] = bos_id
    text_bos[:, 1:] = text[:, :S].to(torch.int32)
    targets = torch.zeros(1, S + 1, dtype=torch.int32, device=dev)
    targets[:, :-1] = text_bos[:, 1:]
    targets[:, -1] = eos_id                                      # equal lengths in the batch: `targets[:, -1] = 0` (:1277)
    with torch.enable_grad():
        out = model.forward(audio_signal, text_bos, a_lengths)
    ctc_out, lm_out = out["final_posteriors_ctc"], out["final_posteriors_lm"]
    N, B = ctc_out.shape[1], ctc_out.shape[0]
    w = float(model.ctc_loss_weight)
    st = torch.cuda.current_stream().cuda_stream
    g_ctc, ctc_show, ctc_bwd = None, 0.0, 0.0
    if w > 0.0:
        ilen = torch.full((B,), N, dtype=torch.int32, device=dev)
        tlen = torch.full((B,), S, dtype=torch.int32, device=dev)
        tg = text[:, :max(S, 1)].to(torch.int32).contiguous() if S else torch.zeros(1, 1, dtype=torch.int32, device=dev)
        loss_ctc, _, g_ctc = ops.ctc_loss(ctc_out.contiguous(), tg, ilen, tlen, ctc_out.shape[-1] - 1, reduction="sum",
                                          grad_scale=w * 100.0 / (N * B), want_grad=backward)
        lc = float(loss_ctc.item())
        ctc_show, ctc_bwd = lc / float(a_lengths.sum()) * 100, lc / (N * B) * 100
    rows, V = S + 1, lm_out.shape[-1]
    logp = ops.log_softmax(lm_out[0].contiguous())
    loss_lm = torch.empty(1, device=dev, dtype=torch.float32)
    row_loss = torch.empty(rows, device=dev, dtype=torch.float32)
    g_lm = torch.empty(rows, V, device=dev, dtype=torch.float32) if backward else None
    check(load().dyn_nll_loss(logp.data_ptr(), targets.data_ptr(), loss_lm.data_ptr(), row_loss.data_ptr(), 0 if g_lm is None else g_lm.data_ptr(),
                              rows, V, -100, (1.0 - w) / (1 * rows), st), "dyn_nll_loss")
    ll = float(loss_lm.item())
    lm_show, lm_bwd = ll / max(1, S), ll / (1 * rows)
    if backward:
        with ops.use_workspace(model._scratch()):
            dh = model._decoder_backward(g_lm)
        gp = g_ctc if g_ctc is not None else torch.zeros_like(ctc_out)
        model.backward(gp, grad_hidden=dh[None])
    return {"loss": ctc_bwd * w + lm_bwd * (1 - w),
            "display_losses": {"loss": ctc_show * w + lm_show * (1 - w), "ctc_loss": ctc_show, "lm_loss": lm_show},
            "ctc_posteriors": ctc_out, "lm_posteriors": lm_out, "length": out["length"]}


def generate_enc_dec(model, audio_signal, **kw):
    """reference call site lib.py:1128: `generate_enc_dec(model, audio_chunk)[0]` -> token ids of the greedy decode.  The
    `sample=4, greedy=False` form is the RL rollout of the grpo / maxrl modes (lib.py:1667-1673): out of scope."""
    if kw.get("sample", 1) != 1 or kw.get("greedy", True) is False:
        raise NotImplementedError("sampled rollouts (RL modes) are out of scope")
    return [torch.tensor(model.generate(audio_signal)["text_sequence"], dtype=torch.long)]


def enc_dec_inference(model, spec, seq_len, overlap, tokenizer, use_tqdm=True):
    """reference lcasr/lib.py:1112-1134: windows without overlap, greedy generate per window, texts joined by one space."""
    assert overlap == 0, 'Overlap not implemented for encoder-decoder model (yet)'
    spec = spec.to(device=model.device, dtype=torch.float32)
    training_data, training_keys = prepare_chunks(spec, seq_len, overlap)
    output_sequences = [None] * len(training_keys)
    idxs = list(range(len(training_keys)))
    for idx in (tqdm(idxs) if use_tqdm else idxs):
        audio_chunk = training_data[training_keys[idx]].contiguous()
        with torch.no_grad():
            output = generate_enc_dec(model, audio_chunk)[0]
        text = tokenizer.decode(output.tolist()).strip()
        print(f'Generated text: {text}')
        output_sequences[idx] = text
    return " ".join(output_sequences).replace('  ', ' ').strip()


def enc_dec_dynamic_eval(args, model, spec, seq_len, overlap, tokenizer, use_tqdm=True, optim=MADGRAD, optimizer_state=None,
                         return_params=False, **kwargs):
    """reference lcasr/lib.py:1475-1732, `training_mode == 'teacher_ce'`: per window — augmented + clean copy (:1574-1576), teacher =
    greedy decode of the clean copy from one no-grad encoder pass (:1580-1588), the teacher filters (:1596-1633), then one
    supervised step on the augmented copy with the teacher's tokens as target: calc_loss_enc_dec -> zero_grad / backward / step
    (:1638-1658); afterwards the whole recording is decoded with the adapted weights (:1714-1722) and the weights restored (:1728-1729)."""
    mode = getattr(args, 'training_mode', 'grpo')
    if mode != 'teacher_ce':
        raise NotImplementedError(f"training_mode {mode!r}: only 'teacher_ce' is implemented (the RL modes grpo / maxrl are out of scope)")
    dropout_emb = args.__dict__.get('dropout_emb', 0.0)
    dropout_post_ff = args.__dict__.get('dropout_post_ff', 0.0)
    dropout_attn = args.__dict__.get('dropout_attn', 0.0)
    for k, v in (('dropout_emb', dropout_emb), ('dropout_post_ff', dropout_post_ff), ('dropout_attn', dropout_attn)):
        if not 0.0 <= float(v) < 1.0:
            raise ValueError(f"{k} must be in [0, 1), got {v}")
    # the reference's knobs on the decoder (lib.py:1519-1525): embedding / post-FFN dropout for the whole call, attention dropout
    # only around the supervised step; draws are counter-based and restart with every call (reproducible: `random_seed`)
    model.language_model_decoder.dropout_emb = dropout_emb
    model.language_model_decoder.ff_out_dropout = dropout_post_ff
    for layer in model.language_model_decoder.layers:
        layer[0].fn.dropout_p = 0
    model.random_seed, model._draws = int(args.__dict__.get('random_seed', 0)), 0
    spec_augment_config = get_specaugment_config_from_args(args)
    print(spec_augment_config)
    lr_args = get_lr_args_from_args(args)
    print(lr_args)
    num_negatives = 1
    device = model.device
    spec = spec.to(device=device, dtype=torch.float32)
    spec_n = spec.shape[-1]
    seq_len = seq_len if seq_len != -1 else args.config['audio_chunking']['size']
    original_flat = model.flat_params.clone()
    optimizer = optim(model.parameters(), **lr_args)
    if optimizer_state is not None:
        optimizer.load_state_dict(optimizer_state)
    augmentation = SpecAugment(**spec_augment_config)
    fixed_masks = args.__dict__.get('spec_augment_fixed_masks', None)
    if seq_len > spec_n:
        seq_len, overlap = spec_n, 0
    else:
        overlap = overlap if overlap != -1 else args.config['audio_chunking']['overlap']
    assert overlap == 0, 'Overlap > 0 not implemented for encoder-decoder model'
    print(f'Using seq_len: {seq_len}')
    ctc_decoder = None
    if args.__dict__.get('teacher_filter_ctc_agreement', False):
        ctc_decoder = GreedyCTCDecoder(tokenizer=tokenizer, blank_id=model.ctc_decoder.num_classes - 1, device=device)
    model.eval()
    training_data, training_keys = prepare_chunks(spec, seq_len, overlap)
    for epoch in range(args.__dict__.get('epochs', 1)):
        print(f'Epoch {epoch + 1} / {args.__dict__.get("epochs", 1)}')
        idxs = list(range(len(training_keys)))
        idxs = random.sample(idxs, len(idxs)) if args.__dict__.get('shuffle', False) else idxs
        for idx in (tqdm(idxs) if use_tqdm else idxs):
            view = training_data[training_keys[idx]][0]                                    # [F, T]
            Fq, u_len = view.shape
            audio_chunk = torch.empty(num_negatives + 1, Fq, u_len, device=device, dtype=torch.float32)
            for b in range(num_negatives + 1):
                audio_chunk[b].copy_(view)
            for b in range(num_negatives):
                masks = fixed_masks[training_keys[idx]] if fixed_masks is not None else augmentation.draw(Fq, u_len)
                if masks[0][0] or masks[1][0]:
                    augmentation.apply(audio_chunk[b], masks, _window_fill_value(audio_chunk[b], augmentation.zero_masking))
            with torch.no_grad():
                encoder_out_for_teacher = model.forward(audio_signal=audio_chunk[-1:].contiguous())
            teacher_pred_tokens = _check_ids(model.generate(audio_chunk[-1:], encoder_states=encoder_out_for_teacher)["text_sequence"],
                                             model.vocab, "teacher prediction")
            teacher_pred = torch.tensor(teacher_pred_tokens, dtype=torch.long, device=device)
            teacher_pred_text = tokenizer.decode(teacher_pred_tokens).strip()
            text_lengths = torch.LongTensor([teacher_pred.shape[-1]])
            acoustic_length = torch.LongTensor([audio_chunk.shape[-1]])
            teacher_mean_max_prob, teacher_mean_entropy, ctc_text, agreement_text = None, None, None, None
            if args.__dict__.get('teacher_filter_low_confidence', False) or args.__dict__.get('teacher_filter_ctc_agreement', False):
                teacher_inputs = torch.zeros(1, teacher_pred.shape[-1] + 1, dtype=torch.int32, device=device)
                teacher_inputs[:, 1:] = teacher_pred.to(torch.int32)
                with torch.no_grad():
                    tf = model.forward(audio_chunk[-1:].contiguous(), teacher_inputs, acoustic_length)
                if args.__dict__.get('teacher_filter_low_confidence', False) and teacher_pred.shape[-1] > 0:
                    lp = ops.log_softmax(tf['final_posteriors_lm'][0, :teacher_pred.shape[-1]].contiguous())
                    _, ent = ops.entropy_grad(lp, 1.0)
                    ids_max, vmax = ops.argmax_rows(lp)
                    teacher_mean_max_prob = float(vmax.exp().mean().item())
                    teacher_mean_entropy = float(ent.mean().item())
                if args.__dict__.get('teacher_filter_ctc_agreement', False) and ctc_decoder is not None:
                    ctc_text = ctc_decoder(tf['final_posteriors_ctc'][0]).strip()
            if args.__dict__.get('teacher_filter_decode_agreement', False):             # lib.py:1620-1627: a second, SAMPLED decode
                agreement_gen = model.generate(audio_chunk[-1:], encoder_states=encoder_out_for_teacher, sample=True,
                                               temperature=args.__dict__.get('teacher_decode_agreement_temperature', 0.7))
                agreement_text = tokenizer.decode(agreement_gen["text_sequence"]).strip()
            print(f'Teacher pred: {teacher_pred_text}')
            skip, reason = should_skip_faulty_teacher_prediction(
                args=args, teacher_pred_tokens=teacher_pred_tokens, teacher_pred_text=teacher_pred_text, spec_frames=audio_chunk.shape[-1],
                agreement_text=agreement_text, teacher_mean_max_prob=teacher_mean_max_prob, teacher_mean_entropy=teacher_mean_entropy, ctc_text=ctc_text)
            if skip:
                print(f'Skipping teacher update: {reason}')
                continue
            for layer in model.language_model_decoder.layers:                            # lib.py:1636-1637
                layer[0].fn.dropout_p = dropout_attn
            model.language_model_decoder.train()   # for dropout
            optimizer.zero_grad()
            out = calc_loss_enc_dec(model=model, audio_signal=audio_chunk[:num_negatives].contiguous(), text_sequence=teacher_pred[None, :],
                                    a_lengths=acoustic_length, t_lengths=text_lengths, tokenizer=tokenizer)
            print(out['loss'], "loss (teacher_ce)")
            optimizer.step()
            model.language_model_decoder.eval()                                          # lib.py:1703-1707
            for layer in model.language_model_decoder.layers:
                layer[0].fn.dropout_p = 0
    model.eval()
    final_out = enc_dec_inference(model=model, spec=spec, seq_len=seq_len, overlap=overlap, tokenizer=tokenizer, use_tqdm=use_tqdm)
    if return_params:
        updated_model_params = [p.clone().detach().cpu() for p in model.parameters()]
    model.flat_params.copy_(original_flat)
    return final_out if not return_params else (final_out, updated_model_params)
