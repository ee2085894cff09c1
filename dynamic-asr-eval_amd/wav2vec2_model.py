"""Wav2Vec2ForCTC on MI355X: the model the reference's wav2vec2 path loads with
`AutoModelForCTC.from_pretrained("facebook/wav2vec2-base-960h")` (reference wav2vec2/lib.py:20-23) and drives with
`model(input_values).logits` (lib.py:163,413), with explicit forward / backward on the HIP kernels only.

Architecture = HF `Wav2Vec2Config()` defaults (= base-960h: 7 strided Conv1d layers of 512 channels, GroupNorm on the
first, feature projection LN + Linear(512 -> 768), grouped positional conv (k = 128, 16 groups, weight-normed) + GELU,
12 post-LN transformer layers with 12 x 64 heads and GELU FFN 3072, lm_head 768 -> 32); `feat_extract_norm="group"`,
`do_stable_layer_norm=False`, eval mode (no dropout, no SpecAugment masking) as in the reference loop.
Parameter NAMES are HF's (state_dict interchange with transformers, which is the oracle in tests); the native HBM layout
of conv kernels is [C_out][kernel][C_in] so each strided Conv1d is one implicit GEMM over overlapping rows of the
channels-last activation (ops.conv1d), converted on load / save."""
import math
import os
from collections import OrderedDict
from types import SimpleNamespace

import torch

from . import ops
from .model import _capture_guard, _no_gc
from .optim import ParamList

DEFAULT_CONFIG = dict(
    conv_dim=(512,) * 7, conv_stride=(5, 2, 2, 2, 2, 2, 2), conv_kernel=(10, 3, 3, 3, 3, 2, 2), hidden_size=768,
    num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, num_conv_pos_embeddings=128,
    num_conv_pos_embedding_groups=16, vocab_size=32, layer_norm_eps=1e-5,
)


def make_config(cfg=None):
    out = dict(DEFAULT_CONFIG)
    if cfg is not None:
        src = cfg if isinstance(cfg, dict) else cfg.__dict__
        for k in DEFAULT_CONFIG:
            if k in src:
                out[k] = tuple(src[k]) if isinstance(src[k], (list, tuple)) else src[k]
        if not isinstance(cfg, dict):
            assert getattr(cfg, "feat_extract_norm", "group") == "group" and not getattr(cfg, "do_stable_layer_norm", False) and \
                not getattr(cfg, "conv_bias", False), "only the base (group-norm, post-LN, bias-free conv) variant is built"
    return out


def _conv_to_native(w):   # HF [Cout, Cin, k] -> native [Cout, k, Cin]
    return w.permute(0, 2, 1).contiguous()


def _conv_to_hf(w):       # native [Cout, k, Cin] -> HF [Cout, Cin, k]
    return w.permute(0, 2, 1).contiguous()


_RELEASED = object()      # Wav2Vec2ForCTC._ctx after a bucket replay whose saved activations are owned by its graphs only


class Wav2Vec2ForCTC:
    def __init__(self, config=None, device="cuda:0"):
        self.cfg = make_config(config)
        c = self.cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ops.DynError("Wav2Vec2ForCTC runs only on the HIP path (device must be cuda)")
        self._defer_arena = None
        H = c["hidden_size"]
        assert H % 256 == 0 and c["conv_dim"][-1] % 256 == 0, "LayerNorm kernels need C % 256 == 0"
        spec, self._conv = [("wav2vec2.masked_spec_embed", (H,), None)], {}
        fe = "wav2vec2.feature_extractor.conv_layers."
        cin = 1
        for i, (co, k) in enumerate(zip(c["conv_dim"], c["conv_kernel"])):
            spec.append((f"{fe}{i}.conv.weight", (co, k, cin), "conv"))
            if i == 0:
                spec += [(f"{fe}0.layer_norm.weight", (co,), None), (f"{fe}0.layer_norm.bias", (co,), None)]
            cin = co
        fp = "wav2vec2.feature_projection."
        spec += [(fp + "layer_norm.weight", (cin,), None), (fp + "layer_norm.bias", (cin,), None),
                 (fp + "projection.weight", (H, cin), None), (fp + "projection.bias", (H,), None)]
        pc = "wav2vec2.encoder.pos_conv_embed.conv."
        K, G = c["num_conv_pos_embeddings"], c["num_conv_pos_embedding_groups"]
        spec += [(pc + "bias", (H,), None), (pc + "parametrizations.weight.original0", (K,), "g"),
                 (pc + "parametrizations.weight.original1", (H, K, H // G), "conv")]
        spec += [("wav2vec2.encoder.layer_norm.weight", (H,), None), ("wav2vec2.encoder.layer_norm.bias", (H,), None)]
        for l in range(c["num_hidden_layers"]):
            p = f"wav2vec2.encoder.layers.{l}."
            # q | k | v weights side by side in the flat buffers (then their biases): one [3H, H] projection instead of three launches
            spec += [(p + f"attention.{n}.weight", (H, H), None) for n in ("q_proj", "k_proj", "v_proj")]
            spec += [(p + f"attention.{n}.bias", (H,), None) for n in ("q_proj", "k_proj", "v_proj")]
            spec += [(p + "attention.out_proj.weight", (H, H), None), (p + "attention.out_proj.bias", (H,), None)]
            spec += [(p + "layer_norm.weight", (H,), None), (p + "layer_norm.bias", (H,), None),
                     (p + "feed_forward.intermediate_dense.weight", (c["intermediate_size"], H), None),
                     (p + "feed_forward.intermediate_dense.bias", (c["intermediate_size"],), None),
                     (p + "feed_forward.output_dense.weight", (H, c["intermediate_size"]), None),
                     (p + "feed_forward.output_dense.bias", (H,), None),
                     (p + "final_layer_norm.weight", (H,), None), (p + "final_layer_norm.bias", (H,), None)]
        spec += [("lm_head.weight", (c["vocab_size"], H), None), ("lm_head.bias", (c["vocab_size"],), None)]
        self.spec = spec
        off, slots = 0, {}
        for name, shape, _ in spec:
            n = math.prod(shape)
            slots[name] = (off, n, shape)
            off += (n + 63) // 64 * 64
        self.flat_params = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.flat_grads = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.P = {n: self.flat_params[o:o + k].view(s) for n, (o, k, s) in slots.items()}
        self.G = {n: self.flat_grads[o:o + k].view(s) for n, (o, k, s) in slots.items()}
        self._kind = {n: kind for n, _, kind in spec}
        # packed views of the attention input projections: [3H, H] weights / [3H] biases (the slots are contiguous when H % 64 == 0)
        self.packed_qkv = H % 64 == 0
        self.Pqkv, self.Gqkv = [], []
        if self.packed_qkv:
            for l in range(c["num_hidden_layers"]):
                ow = slots[f"wav2vec2.encoder.layers.{l}.attention.q_proj.weight"][0]
                ob = slots[f"wav2vec2.encoder.layers.{l}.attention.q_proj.bias"][0]
                assert slots[f"wav2vec2.encoder.layers.{l}.attention.v_proj.weight"][0] == ow + 2 * H * H
                assert slots[f"wav2vec2.encoder.layers.{l}.attention.v_proj.bias"][0] == ob + 2 * H
                self.Pqkv.append((self.flat_params[ow:ow + 3 * H * H].view(3 * H, H), self.flat_params[ob:ob + 3 * H]))
                self.Gqkv.append((self.flat_grads[ow:ow + 3 * H * H].view(3 * H, H), self.flat_grads[ob:ob + 3 * H]))
        self.frozen = set()
        self._ctx = None
        self._wq = None
        self.grouped_wgrad = os.environ.get("DYN_GROUPED_WGRAD", "1") != "0"     # A/B switch: 0 = one launch per weight gradient
        self.config = SimpleNamespace(**c)
        # hipGraph replay over LENGTH BUCKETS (see forward()): off unless a loop turns it on (wav2vec2_lib: args.use_graphs)
        self.use_graphs = False
        self.bucket_frames = 32                  # an utterance of T' frames runs in the bucket of ceil(T' / 32) * 32 frames (0.64 s of audio)
        self.graph_after = 2                     # a bucket is captured the n-th time it is seen; before that the utterance runs unpadded, eagerly
        self.graph_budget_bytes = 64 << 30       # device memory the buckets' graphs may hold (their shared pool + static buffers); beyond it all are dropped
        self._graphs = OrderedDict()             # bucket key -> {"graph", "in", "out", "ctx", "bwd": {...}, "bytes"}
        self._pool = None                        # ONE private memory pool for all buckets of this model (see _forward_bucketed)
        self._seen = {}
        self._valid = None                       # device int32 [n_conv]: valid frames after every conv layer of the utterance in flight
        self._vl = None                          # = self._valid while a bucketed launch sequence is being issued / captured, else None
        self._ws = None
        self._ctx_static, self._ctx_key = False, None

    # ------------------------------------------------------------------ nn.Module-like surface
    def named_parameters(self):
        return [(n, self.P[n]) for n, _, _ in self.spec]

    def parameters(self):
        pl = ParamList(self.P[n] for n, _, _ in self.spec)
        pl.flat_params, pl.flat_grads = self.flat_params, self.flat_grads
        return pl

    def grads_hf(self):
        """{name: gradient in HF layout} (tests)."""
        return {n: self._to_hf(n, self.G[n]) for n, _, _ in self.spec}

    def _to_hf(self, name, t):
        kind = self._kind[name]
        if kind == "conv":
            return _conv_to_hf(t)
        if kind == "g":
            return t.reshape(1, 1, -1)
        return t

    def state_dict(self):
        return {n: self._to_hf(n, p).detach().clone() for n, p in self.named_parameters()}

    def load_state_dict(self, sd, strict=True):
        missing = [n for n, _, _ in self.spec if n not in sd]
        if strict and missing:
            raise KeyError(f"missing {missing[:4]}…")
        for n, _, kind in self.spec:
            if n not in sd:
                continue
            t = sd[n].to(torch.float32)
            if kind == "conv":
                t = _conv_to_native(t)
            self.P[n].copy_(t.reshape(self.P[n].shape).to(self.device))
        return SimpleNamespace(missing_keys=missing, unexpected_keys=[k for k in sd if k not in self.P])

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def to(self, device):
        return self

    def modules(self):
        return []

    def zero_grad(self):
        self.flat_grads.zero_()

    # ------------------------------------------------------------------ forward
    def __call__(self, input_values, **kw):
        return self.forward(input_values)

    def conv_lengths(self, L):
        """Frames after every layer of the feature extractor for L input samples."""
        out = []
        for k, st in zip(self.cfg["conv_kernel"], self.cfg["conv_stride"]):
            L = (L - k) // st + 1
            out.append(L)
        return out

    def samples_for_frames(self, T):
        """The smallest number of input samples that gives T output frames."""
        L = T
        for k, st in zip(reversed(self.cfg["conv_kernel"]), reversed(self.cfg["conv_stride"])):
            L = (L - 1) * st + k
        return L

    def _scratch(self):
        if self._ws is None:        # the model's own scratch (ops.use_workspace): a stream-keyed buffer is wrong inside a capture
            self._ws = torch.empty(ops.WORKSPACE_BYTES, dtype=torch.uint8, device=self.device)
            ops.counters(self._ws)
            self._valid = torch.zeros(len(self.cfg["conv_kernel"]), dtype=torch.int32, device=self.device)
        return self._ws

    def forward(self, input_values):
        """input_values [B, L] float32 on the device (already zero-mean / unit-variance) -> SimpleNamespace(logits [B, T', V], frames).

        With `use_graphs` the launch sequence of a LENGTH BUCKET is captured once as a hipGraph and replayed for every utterance that
        falls into it (the per-utterance loop, reference wav2vec2/lib.py:293-462, is launch-bound — ~500 launches
        of work for ~10 ms of audio model time, and every utterance has its own length).  The waveform is zero-padded to the bucket's
        sample count; `logits` then has the BUCKET's frame count and `frames` says how many of them belong to the utterance.  The
        utterance's own frame counts live in HBM (`_valid`) and are read by the kernels when the graph runs, so that the valid frames
        are those of the unpadded run: the first layer's GroupNorm takes its statistics over the valid frames only
        (dyn_colnorm_fwd_len), the frames past the end are zeroed before the positional conv (dyn_mask_rows: what its zero padding
        holds there), and attention masks the keys past the end (dyn_softmax_fwd_len).  Everything else is per frame, or (the strided
        convs) looks only backwards from a valid frame.  The backward gets exact zeros on the padded frames (CTC gives them zero
        gradient, every per-frame kernel maps 0 to 0, the masked softmax cuts the attention path, dyn_mask_rows / dyn_colnorm_bwd_len
        cut the other two), so every weight gradient sums the same terms plus zeros: equal to the unpadded run up to the summation
        order of a longer K (tests/test_wav2vec2_gpu.py)."""
        x = input_values
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
            raise ops.DynError("input_values must be a float32 CUDA tensor [B, L]")
        with ops.use_workspace(self._scratch()):
            if self.use_graphs and not ops.GEMM_PROFILE_EAGER():
                return self._forward_bucketed(x)
            self._vl, self._ctx_static = None, False
            return self._forward_eager(x)

    def _bucket(self, L):
        T = self.conv_lengths(L)[-1]
        q = max(1, int(self.bucket_frames))
        Tb = -(-T // q) * q
        return T, Tb, self.samples_for_frames(Tb + 1) - 1       # every L with <= Tb frames fits; the bucket itself has exactly Tb frames

    def graph_bytes(self):
        return sum(e["bytes"] for e in self._graphs.values())

    def drop_graphs(self):
        """Forget every captured bucket (never inside a capture: destroying a graph there aborts the process).  The shared pool goes back to the
        allocator once its last graph is gone."""
        if self._graphs:
            torch.cuda.synchronize(self.device)             # no replay may still be in flight when its graph is destroyed
        for ent in self._graphs.values():
            ent.clear()
        self._graphs.clear()
        self._pool = None
        if self._ctx is _RELEASED:
            self._ctx = None
        self._ctx_static = False

    def _forward_bucketed(self, x):
        B, L = x.shape
        need = self.samples_for_frames(1)
        if L < need:
            raise ops.DynError(f"input of {L} samples is shorter than the feature extractor's receptive field ({need} samples)")
        T, Tb, Lb = self._bucket(L)
        key = (B, Lb, torch.is_grad_enabled())
        ent = self._graphs.get(key)
        if ent is None:
            n = self._seen[key] = self._seen.get(key, 0) + 1
            if n < self.graph_after:
                self._vl, self._ctx_static = None, False
                return self._forward_eager(x)
            if self.graph_bytes() > self.graph_budget_bytes:
                self.drop_graphs()
            # ONE private pool for every bucket of this model, and no Python reference to a bucket's saved activations once its backward graph
            # exists (_backward_graphed): the next bucket's capture then reuses that memory, so the pool grows to the LARGEST bucket's needs,
            # not to their sum (15 buckets of one 5-min talk held 26 GiB with a pool each).  Safe because the graphs of one model replay
            # strictly as forward(b), backward(b) pairs on one stream: a bucket's activations only have to survive from its forward replay to
            # its own backward replay, and no other graph of this model runs in between.  (Not the interleaved case of model.py::_graph_pool.)
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            r0 = torch.cuda.memory_reserved(self.device)
            static_in = torch.zeros(B, Lb, device=self.device, dtype=torch.float32)
            graph = torch.cuda.CUDAGraph()
            prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
            self._vl = self._valid
            try:
                with _no_gc(), torch.cuda.graph(graph, pool=self._pool, capture_error_mode="thread_local"), _capture_guard():
                    out = self._forward_eager(static_in)
            finally:
                ops.GEMM_PROFILE, self._vl = prof, None
            ent = {"graph": graph, "in": static_in, "out": out.logits, "ctx": self._ctx, "bwd": {}, "len": 0,
                   "bytes": max(0, torch.cuda.memory_reserved(self.device) - r0)}
            self._graphs[key] = ent
        self._graphs.move_to_end(key)
        ent["in"][:, :L].copy_(x)
        if L < ent["len"]:
            ent["in"][:, L:ent["len"]].zero_()
        ent["len"] = L                                   # (beyond the previous utterance's end the buffer is still zero)
        cl = self.conv_lengths(L)
        self._valid[0:1].fill_(cl[0])                    # (fills, not a pageable upload: that would block the host on this stream and stall the
        self._valid[-1:].fill_(cl[-1])                   #  other chains of wav2vec2_lib.dynamic_eval_su_many)
        ent["graph"].replay()
        self._ctx, self._ctx_static, self._ctx_key = (ent["ctx"] if ent["ctx"] is not None else _RELEASED), True, key
        return SimpleNamespace(logits=ent["out"], frames=T)

    def _forward_eager(self, x):
        c, P = self.cfg, self.P
        vl = self._vl
        v0 = vl[0:1] if vl is not None else None             # valid frames after the first conv (its GroupNorm runs over time)
        vT = vl[-1:] if vl is not None else None             # valid frames of the encoder
        save = torch.is_grad_enabled()
        ctx = {"conv": []} if save else None
        B, L = x.shape
        need = 1
        for k, st in zip(reversed(c["conv_kernel"]), reversed(c["conv_stride"])):   # receptive field of the conv stack (400 samples)
            need = (need - 1) * st + k
        if L < need:
            raise ops.DynError(f"input of {L} samples is shorter than the feature extractor's receptive field ({need} samples)")
        fe = "wav2vec2.feature_extractor.conv_layers."
        a = x.contiguous().view(B, L, 1)
        for i, (k, s) in enumerate(zip(c["conv_kernel"], c["conv_stride"])):
            w = P[f"{fe}{i}.conv.weight"]
            z = ops.conv1d(a, w.view(w.shape[0], -1), k, s)
            if i == 0:
                n0, mean, rstd = ops.colnorm(z, P[fe + "0.layer_norm.weight"], P[fe + "0.layer_norm.bias"], c["layer_norm_eps"], valid=v0)
                act = ops.gelu(n0)
                if save:
                    ctx["conv"].append((a, z, (mean, rstd, n0)))
            else:
                act = ops.gelu(z)
                if save:
                    ctx["conv"].append((a, z, None))
            a = act
        T = a.shape[1]
        fp = "wav2vec2.feature_projection."
        n, mean, rstd = ops.layernorm(a, P[fp + "layer_norm.weight"], P[fp + "layer_norm.bias"], c["layer_norm_eps"])
        h = ops.linear(n, P[fp + "projection.weight"], P[fp + "projection.bias"])
        if vT is not None:
            ops.mask_rows(h, vT)                                    # the positional conv must see zeros past the utterance's last frame
        if save:
            ctx["proj"] = (a, mean, rstd, n)
        # positional conv embedding (grouped, weight-normed), GELU, residual, LayerNorm
        pc = "wav2vec2.encoder.pos_conv_embed.conv."
        H, K, G = c["hidden_size"], c["num_conv_pos_embeddings"], c["num_conv_pos_embedding_groups"]
        cg, pad = H // G, K // 2
        v, g = P[pc + "parametrizations.weight.original1"], P[pc + "parametrizations.weight.original0"]
        w = ops.weight_norm(v, g)                                   # [H, K, cg]
        xg = ops.group_pack(h, G, pad)                              # [B, G, T + 2 pad, cg]
        Tp = T + 2 * pad
        yg = torch.empty(B, G, T, cg, device=h.device, dtype=torch.float32)
        ops.gemm(xg, w, yg, trans_b=True, M=T, N=cg, K=K * cg, lda=cg, ldb=K * cg, ldc=cg, nb1=B, nb2=G,
                 sa=(G * Tp * cg, Tp * cg), sb=(0, cg * K * cg), sc=(G * T * cg, T * cg))
        pre = ops.group_unpack(yg, P[pc + "bias"], T, H)
        pos = ops.gelu(pre)
        hs = h.clone() if save else h
        ops.axpby(pos, hs, 1.0, 1.0)
        h2, mean, rstd = ops.layernorm(hs, P["wav2vec2.encoder.layer_norm.weight"], P["wav2vec2.encoder.layer_norm.bias"], c["layer_norm_eps"])
        if save:
            ctx["pos"] = (xg, w, pre, hs, mean, rstd)
            ctx["layers"] = []
        h = h2
        nh = c["num_attention_heads"]
        D = H // nh
        for l in range(c["num_hidden_layers"]):
            p = f"wav2vec2.encoder.layers.{l}."
            qkv = torch.empty(B, T, 3 * H, device=h.device, dtype=torch.float32)
            if self.packed_qkv:
                ops.linear(h, self.Pqkv[l][0], self.Pqkv[l][1], out=qkv)
            for j, nm in enumerate(() if self.packed_qkv else ("q_proj", "k_proj", "v_proj")):
                ops.gemm(h, P[p + f"attention.{nm}.weight"], qkv, trans_b=True, M=B * T, N=H, K=H, lda=H, ldb=H, ldc=3 * H,
                         c_off=j * H, bias=P[p + f"attention.{nm}.bias"])
            S = torch.empty(B, nh, T, T, device=h.device, dtype=torch.float32)
            ops.gemm(qkv, qkv, S, trans_b=True, M=T, N=T, K=D, lda=3 * H, ldb=3 * H, ldc=T, nb1=B, nb2=nh,
                     sa=(T * 3 * H, D), sb=(T * 3 * H, D), sc=(nh * T * T, T * T), b_off=H, alpha=D ** -0.5)
            ops.softmax(S, out=S, valid=vT)
            O = torch.empty(B, T, H, device=h.device, dtype=torch.float32)
            ops.gemm(S, qkv, O, M=T, N=D, K=T, lda=T, ldb=3 * H, ldc=H, nb1=B, nb2=nh, sa=(nh * T * T, T * T),
                     sb=(T * 3 * H, D), sc=(T * H, D), b_off=2 * H)
            r1 = h.clone() if save else h
            ops.linear(O, P[p + "attention.out_proj.weight"], P[p + "attention.out_proj.bias"], out=r1, beta=1.0)
            h1, m1, s1 = ops.layernorm(r1, P[p + "layer_norm.weight"], P[p + "layer_norm.bias"], c["layer_norm_eps"])
            u = ops.linear(h1, P[p + "feed_forward.intermediate_dense.weight"], P[p + "feed_forward.intermediate_dense.bias"])
            ga = ops.gelu(u)
            r2 = h1.clone() if save else h1
            ops.linear(ga, P[p + "feed_forward.output_dense.weight"], P[p + "feed_forward.output_dense.bias"], out=r2, beta=1.0)
            h2, m2, s2 = ops.layernorm(r2, P[p + "final_layer_norm.weight"], P[p + "final_layer_norm.bias"], c["layer_norm_eps"])
            if save:
                ctx["layers"].append((h, qkv, S, O, r1, m1, s1, h1, u, ga, r2, m2, s2))
            h = h2
        logits = ops.linear(h, P["lm_head.weight"], P["lm_head.bias"])
        if save:
            ctx["head"] = h
            ctx["dims"] = (B, L, T)
            ctx["valid"] = (v0, vT)
        self._ctx = ctx
        return SimpleNamespace(logits=logits, frames=T)

    # ------------------------------------------------------------------ backward
    def _wgrad(self, dy, x, dw, db):
        """dw += dy^T x, db += column sums of dy: queued for the ONE grouped launch that ends the backward (the products of a short
        utterance are 36 - 144 tiles each: far too few to fill the chip alone), or at once.  Queued operands must stay unmodified."""
        dy = dy.contiguous()
        if self._wq is not None and ops.wgrad_groupable(dy, x, dw):
            self._wq.append(ops.wgrad_desc(dy, x, dw, beta=1.0, colsum=db, colsum_beta=1.0))
            return
        ops.linear_wgrad(dy, x, dw, beta=1.0)
        if db is not None:
            ops.colsum(dy, db, beta=1.0)

    def _lin_bwd(self, dy, x, wname, bname, need_dx=True):
        self._wgrad(dy, x, self.G[wname], self.G[bname] if bname is not None else None)
        return ops.linear_dgrad(dy, self.P[wname]) if need_dx else None

    def backward(self, grad_logits, n_active=None):
        """Accumulates dL/dparam into flat_grads given dL/dlogits [nb, T', V] for the first nb samples of the batch.  The launch-bound
        column reductions of the norm / bias gradients run as one batched launch at the end (ops.reduce_defer; bit-identical)."""
        if self._defer_arena is None and os.environ.get("DYN_DEFER_REDUCE", "1") != "0":
            self._defer_arena = torch.empty(ops.DEFER_ARENA_BYTES, dtype=torch.uint8, device=self.device)
        with ops.use_workspace(self._scratch()):
            if self._ctx_static and self.use_graphs and self._ctx is not None:
                return self._backward_graphed(grad_logits, n_active)
            self._backward_eager(grad_logits, n_active)

    def _backward_graphed(self, grad_logits, n_active):
        """Replay (first use: capture) of the backward launch sequence of the bucket whose forward graph produced the saved activations."""
        ent = self._graphs[self._ctx_key]
        key = (tuple(grad_logits.shape), n_active, frozenset(self.frozen), self.grouped_wgrad)
        b = ent["bwd"].get(key)
        if b is None and ent["ctx"] is None:
            # another backward variant (other frozen set / active copies) after the bucket's activations were released: recompute them eagerly from
            # the bucket's input buffer (the utterance and its frame counts are still there) and run this backward eagerly
            self._vl = self._valid
            try:
                with torch.enable_grad():
                    self._forward_eager(ent["in"])
            finally:
                self._vl = None
            self._ctx_static = False
            return self._backward_eager(grad_logits, n_active)
        if b is None:
            self._ctx = ent["ctx"]
            r0 = torch.cuda.memory_reserved(self.device)
            static_g = grad_logits.contiguous().clone()
            graph = torch.cuda.CUDAGraph()
            prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
            try:
                with _no_gc(), torch.cuda.graph(graph, pool=self._pool, capture_error_mode="thread_local"), _capture_guard():
                    self._backward_eager(static_g, n_active)
            finally:
                ops.GEMM_PROFILE = prof
            b = ent["bwd"][key] = {"graph": graph, "g": static_g}
            ent["bytes"] += max(0, torch.cuda.memory_reserved(self.device) - r0)
            ent["ctx"] = None                                    # released: both graphs hold the addresses, nobody needs the tensors
        b["g"].copy_(grad_logits)
        b["graph"].replay()
        self._ctx = None

    def _backward_eager(self, grad_logits, n_active):
        with ops.reduce_defer(self._defer_arena):
            self._backward_body(grad_logits, n_active)
        for name in self.frozen:                    # after the deferred reductions have been queued: they write gradients too
            for n, _, _ in self.spec:
                if n.startswith(name):
                    self.G[n].zero_()

    def _backward_body(self, grad_logits, n_active=None):
        self._wq = [] if self.grouped_wgrad else None
        try:
            self._backward_layers(grad_logits, n_active)
            if self._wq:
                ops.gemm_grouped(self._wq)      # every linear layer's weight gradient (+ bias sums) of this backward: one launch
        finally:
            self._wq = None

    def _backward_layers(self, grad_logits, n_active=None):
        ctx = self._ctx
        if ctx is None or ctx is _RELEASED:
            raise ops.DynError("backward() without a grad-mode forward (or use_graphs switched off between a bucketed forward and its backward)")
        if self._ctx_static:                        # a graph-owned context must survive the backward (entries are dropped below as they are used)
            ctx = dict(ctx)
            ctx["layers"], ctx["conv"] = list(ctx["layers"]), list(ctx["conv"])
        v0, vT = ctx["valid"]
        c, P, G = self.cfg, self.P, self.G
        B, L, T = ctx["dims"]
        nb = B if n_active is None else int(n_active)
        assert grad_logits.shape[0] == nb

        def cut(t):
            if isinstance(t, tuple):
                return tuple(cut(x) for x in t)
            if t is None or nb == B:
                return t
            if t.dim() == 1:
                return t[:nb * (t.shape[0] // B)]
            return t[:nb]

        H, nh = c["hidden_size"], c["num_attention_heads"]
        D = H // nh
        eps = c["layer_norm_eps"]
        h_last = cut(ctx["head"])
        dh = self._lin_bwd(grad_logits.contiguous(), h_last, "lm_head.weight", "lm_head.bias")
        for l in reversed(range(c["num_hidden_layers"])):
            p = f"wav2vec2.encoder.layers.{l}."
            h, qkv, S, O, r1, m1, s1, h1, u, ga, r2, m2, s2 = cut(ctx["layers"][l])
            dr2 = torch.empty_like(dh)
            ops.layernorm_bwd(r2, P[p + "final_layer_norm.weight"], m2, s2, dh, dr2, G[p + "final_layer_norm.weight"],
                              G[p + "final_layer_norm.bias"], dx_beta=0.0)
            dga = self._lin_bwd(dr2, ga, p + "feed_forward.output_dense.weight", p + "feed_forward.output_dense.bias")
            du = ops.gelu_bwd(u, dga, out=dga)
            dh1 = self._lin_bwd(du, h1, p + "feed_forward.intermediate_dense.weight", p + "feed_forward.intermediate_dense.bias")
            ops.axpby(dr2, dh1, 1.0, 1.0)                            # residual: h1 feeds both the FFN and r2
            dr1 = torch.empty_like(dh1)
            ops.layernorm_bwd(r1, P[p + "layer_norm.weight"], m1, s1, dh1, dr1, G[p + "layer_norm.weight"], G[p + "layer_norm.bias"],
                              dx_beta=0.0)
            dO = self._lin_bwd(dr1, O, p + "attention.out_proj.weight", p + "attention.out_proj.bias")
            dqkv = torch.empty_like(qkv)
            sS, sQ, sO = (nh * T * T, T * T), (T * 3 * H, D), (T * H, D)
            ops.gemm(S, dO, dqkv, trans_a=True, M=T, N=D, K=T, lda=T, ldb=H, ldc=3 * H, nb1=nb, nb2=nh, sa=sS, sb=sO, sc=sQ, c_off=2 * H)
            dP = torch.empty_like(S)
            ops.gemm(dO, qkv, dP, trans_b=True, M=T, N=T, K=D, lda=H, ldb=3 * H, ldc=T, nb1=nb, nb2=nh, sa=sO, sb=sQ, sc=sS, b_off=2 * H)
            ops.softmax_bwd(S, dP, out=dP, scale=1.0)
            sc = D ** -0.5
            ops.gemm(dP, qkv, dqkv, M=T, N=D, K=T, lda=T, ldb=3 * H, ldc=3 * H, nb1=nb, nb2=nh, sa=sS, sb=sQ, sc=sQ, b_off=H, c_off=0, alpha=sc)
            ops.gemm(dP, qkv, dqkv, trans_a=True, M=T, N=D, K=T, lda=T, ldb=3 * H, ldc=3 * H, nb1=nb, nb2=nh, sa=sS, sb=sQ, sc=sQ,
                     b_off=0, c_off=H, alpha=sc)
            M = nb * T
            if self.packed_qkv:                                      # the three projections as one [3H, H] product each way
                self._wgrad(dqkv.view(M, 3 * H), h.view(M, H), self.Gqkv[l][0], self.Gqkv[l][1])
                dh_in = torch.empty_like(dr1)                        # dr1 (residual path) may be a queued operand: not accumulated in place
                ops.gemm(dqkv, self.Pqkv[l][0], dh_in, M=M, N=H, K=3 * H, lda=3 * H, ldb=H, ldc=H, beta=1.0, c_in=dr1)
            else:
                dh_in = dr1.clone() if self._wq is not None else dr1
            for j, nm in enumerate(() if self.packed_qkv else ("q_proj", "k_proj", "v_proj")):
                # dW += dqkv_j^T h ; db += colsum ; dh += dqkv_j W      (dqkv_j is a strided [M, H] slice, lda = 3H)
                ops.gemm(dqkv, h, G[p + f"attention.{nm}.weight"], trans_a=True, M=H, N=H, K=M, lda=3 * H, ldb=H, ldc=H, a_off=j * H, beta=1.0)
                dj = dqkv.view(M, 3, H)[:, j, :].contiguous()
                ops.colsum(dj, G[p + f"attention.{nm}.bias"], beta=1.0)
                ops.gemm(dqkv, P[p + f"attention.{nm}.weight"], dh_in, M=M, N=H, K=H, lda=3 * H, ldb=H, ldc=H, a_off=j * H, beta=1.0)
            dh = dh_in
            ctx["layers"][l] = None
        # encoder LayerNorm, positional conv
        w = ctx["pos"][1]                                            # the normalised weight is not batch-indexed
        xg, _, pre, hs, mean, rstd = cut(ctx["pos"])
        pc = "wav2vec2.encoder.pos_conv_embed.conv."
        K, Gn = c["num_conv_pos_embeddings"], c["num_conv_pos_embedding_groups"]
        cg, pad = H // Gn, K // 2
        Tp = T + 2 * pad
        dhs = torch.empty_like(dh)
        ops.layernorm_bwd(hs, P["wav2vec2.encoder.layer_norm.weight"], mean, rstd, dh, dhs, G["wav2vec2.encoder.layer_norm.weight"],
                          G["wav2vec2.encoder.layer_norm.bias"], dx_beta=0.0)
        dpre = ops.gelu_bwd(pre, dhs)
        ops.colsum(dpre, G[pc + "bias"], beta=1.0)
        dyg = ops.group_pack_grad(dpre, Gn, T)                       # [nb, G, T, cg]
        dw = torch.zeros_like(w)
        for b in range(nb):                                          # dw[g] += dyg[b, g]^T rows(xg[b, g])
            ops.gemm(dyg, xg, dw, trans_a=True, M=cg, N=K * cg, K=T, lda=cg, ldb=cg, ldc=K * cg, nb1=1, nb2=Gn,
                     sa=(0, T * cg), sb=(0, Tp * cg), sc=(0, cg * K * cg), a_off=b * Gn * T * cg, b_off=b * Gn * Tp * cg, beta=1.0)
        v, g = P[pc + "parametrizations.weight.original1"], P[pc + "parametrizations.weight.original0"]
        ops.weight_norm_bwd(v, g, dw, G[pc + "parametrizations.weight.original1"], G[pc + "parametrizations.weight.original0"], beta=1.0)
        dA = torch.empty(nb, Gn, T, K * cg, device=dh.device, dtype=torch.float32)
        ops.gemm(dyg, w, dA, M=T, N=K * cg, K=cg, lda=cg, ldb=K * cg, ldc=K * cg, nb1=nb, nb2=Gn, sa=(Gn * T * cg, T * cg),
                 sb=(0, cg * K * cg), sc=(Gn * T * K * cg, T * K * cg))
        dxg = torch.empty(nb * Gn, Tp, cg, device=dh.device, dtype=torch.float32)
        from ._lib import check, load
        check(load().dyn_col2im_1d(dA.data_ptr(), dxg.data_ptr(), nb * Gn, Tp, T, cg, K, 1, torch.cuda.current_stream().cuda_stream),
              "dyn_col2im_1d")
        ops.group_unpack_grad(dxg.view(nb, Gn, Tp, cg), dhs, pad, beta=1.0)   # dh (pre-pos) = dhs (residual) + pos-conv path
        if vT is not None:
            ops.mask_rows(dhs, vT)                                   # the windows of the last valid frames reach into the zeroed tail: no gradient there
        # feature projection
        a, mean, rstd, n = cut(ctx["proj"])
        fp = "wav2vec2.feature_projection."
        dn = self._lin_bwd(dhs, n, fp + "projection.weight", fp + "projection.bias")
        da = torch.empty_like(dn)
        ops.layernorm_bwd(a, P[fp + "layer_norm.weight"], mean, rstd, dn, da, G[fp + "layer_norm.weight"], G[fp + "layer_norm.bias"], dx_beta=0.0)
        # conv feature extractor
        fe = "wav2vec2.feature_extractor.conv_layers."
        for i in reversed(range(len(c["conv_kernel"]))):
            k, s = c["conv_kernel"][i], c["conv_stride"][i]
            a_in, z, norm = cut(ctx["conv"][i])
            wname = f"{fe}{i}.conv.weight"
            wmat = P[wname].view(P[wname].shape[0], -1)
            if i == 0:
                mean, rstd, n0 = norm
                dn0 = ops.gelu_bwd(n0, da)
                dz = ops.colnorm_bwd(z, P[fe + "0.layer_norm.weight"], mean, rstd, dn0, G[fe + "0.layer_norm.weight"], G[fe + "0.layer_norm.bias"],
                                     valid=v0)
            else:
                dz = ops.gelu_bwd(z, da)
            ops.conv1d_wgrad(a_in, dz, G[wname].view(wmat.shape), k, s, beta=1.0)
            if i > 0:
                da = ops.conv1d_dgrad(dz, wmat, a_in.shape[1], a_in.shape[2], k, s)
            ctx["conv"][i] = None
        self._ctx = None
