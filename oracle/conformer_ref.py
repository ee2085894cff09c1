"""ORACLE (test infrastructure only — never imported by the product path; used by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg as the checker / CPU baseline).

Plain PyTorch fp32 CPU restatement of the acoustic model the reference drives through
`model(audio_signal=...)['final_posteriors']` (reference lcasr/lib.py:550,559,603) with the attribute surface it
touches: `.decoder.num_classes` (lib.py:492), `.device` (lib.py:549), `.parameters()` (lib.py:482,494),
`.subsampling` / `.layers` / `.decoder` (lib.py:164,179-180).

The model class itself (`SCConformerXL`, reference earnings_finetune/lcasr160rb1.yaml:71) lives in the un-vendored,
un-pinned package `lcasr` (upstream repo long-context-asr), which is absent from /root/reference and from this
container.  Every hyper-parameter the reference DOES hold is honoured (yaml:1-29: feat_in 80, n_layers 6, d_model 768,
n_heads 6, head_dim 128, dw_striding x8 subsampling with 256 channels and SiLU, conv kernel 9, rotary base 1.5e6,
self-conditioning, decoder norm, layer_norm default norm, no FF bias); what the yaml does not pin is DEFINED here and
shared with the HIP implementation: FF multiplier 4, macaron pre-norm block order FF/2 -> MHSA -> conv -> FF/2 ->
LayerNorm, conv-module norm (rms_norm | layer_norm | batch_renorm in eval mode), GLU/SiLU placement, NeMo-style
dw_striding internals, channels-last flatten order of the subsampling output, rotate-half rotary convention, and
self-conditioning as x += reproj(softmax(decoder(x))) after every block but the last.
PARITY UNPINNED against upstream weights/outputs: the reference holds no checkpoint, fixture or test for this model.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

DEFAULT_CONFIG = dict(
    feat_in=80, n_layers=6, d_model=768, n_heads=6, head_dim=128, ff_mult=4, subsampling_factor=8,
    subsampling_conv_channels=256, conv_kernel_size=9, conv_norm="rms_norm", rotary_base_freq=1500000.0,
    self_conditioning=True, norm_eps=1e-5,
)


def make_config(**over):
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(over)
    return cfg


def rotary_tables(T, D, base):
    """cos/sin [T, D/2] computed in float64 and rounded once to fp32 (shared by oracle and HIP path)."""
    inv = 1.0 / (float(base) ** (torch.arange(0, D, 2, dtype=torch.float64) / D))
    ang = torch.arange(T, dtype=torch.float64)[:, None] * inv[None]
    return ang.cos().float(), ang.sin().float()


class _Lin(nn.Module):
    def __init__(self, i, o, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.empty(o)) if bias else None

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class _Norm(nn.Module):
    def __init__(self, d, kind="layer_norm", eps=1e-5):
        super().__init__()
        self.kind, self.eps = kind, eps
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d)) if kind != "rms_norm" else None
        if kind == "batch_renorm":  # eval mode: running statistics, no batch statistics (reference lib.py:525)
            self.register_buffer("running_mean", torch.zeros(d))
            self.register_buffer("running_var", torch.ones(d))

    def forward(self, x):
        if self.kind == "layer_norm":
            return F.layer_norm(x, x.shape[-1:], self.weight, self.bias, self.eps)
        if self.kind == "rms_norm":
            return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps) * self.weight
        return (x - self.running_mean) * torch.rsqrt(self.running_var + self.eps) * self.weight + self.bias


class _DW(nn.Module):
    def __init__(self, c, *k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c, *k))
        self.bias = nn.Parameter(torch.empty(c))


class Subsampling(nn.Module):
    """dw_striding x8: conv3x3/s2(1->C) | SiLU, dw3x3/s2 + pw1x1 | SiLU, dw3x3/s2 + pw1x1 | SiLU, Linear(F/8*C -> d)."""

    def __init__(self, cfg):
        super().__init__()
        C, d = cfg["subsampling_conv_channels"], cfg["d_model"]
        fo = cfg["feat_in"]
        for _ in range(3):
            fo = (fo - 1) // 2 + 1
        self.conv1 = _DW(C, 3, 3)
        self.dw2, self.pw2 = _DW(C, 3, 3), _Lin(C, C)
        self.dw3, self.pw3 = _DW(C, 3, 3), _Lin(C, C)
        self.out = _Lin(fo * C, d)

    def forward(self, x):  # x [B, F, T]
        C = self.conv1.weight.shape[0]
        h = x.transpose(1, 2).unsqueeze(1)  # [B, 1, T, F]
        z = F.conv2d(h, self.conv1.weight.unsqueeze(1), self.conv1.bias, stride=2, padding=1)
        for dw, pw in ((self.dw2, self.pw2), (self.dw3, self.pw3)):
            u = F.conv2d(F.silu(z), dw.weight.unsqueeze(1), dw.bias, stride=2, padding=1, groups=C)
            z = F.conv2d(u, pw.weight[:, :, None, None], pw.bias)
        a = F.silu(z).permute(0, 2, 3, 1)  # [B, T', F', C]  (flatten order (f, c))
        return self.out(a.reshape(a.shape[0], a.shape[1], -1))


class FeedForward(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg["d_model"]
        self.norm = _Norm(d, "layer_norm", cfg["norm_eps"])
        self.w1 = _Lin(d, d * cfg["ff_mult"], bias=False)
        self.w2 = _Lin(d * cfg["ff_mult"], d, bias=False)

    def forward(self, x):
        return self.w2(F.silu(self.w1(self.norm(x))))


class Attention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d, self.H, self.D = cfg["d_model"], cfg["n_heads"], cfg["head_dim"]
        self.base = cfg["rotary_base_freq"]
        self.norm = _Norm(d, "layer_norm", cfg["norm_eps"])
        self.qkv = _Lin(d, 3 * self.H * self.D)
        self.out = _Lin(self.H * self.D, d)

    def forward(self, x):
        B, T, _ = x.shape
        H, D = self.H, self.D
        qkv = self.qkv(self.norm(x)).view(B, T, 3, H, D)
        cos, sin = rotary_tables(T, D, self.base)
        c, s = cos[None, :, None], sin[None, :, None]

        def rot(t):
            t1, t2 = t[..., : D // 2], t[..., D // 2:]
            return torch.cat([t1 * c - t2 * s, t2 * c + t1 * s], -1)

        q, k, v = rot(qkv[:, :, 0]), rot(qkv[:, :, 1]), qkv[:, :, 2]
        q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))  # [B, H, T, D]
        p = torch.softmax((q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(D)), -1)
        o = (p @ v).permute(0, 2, 1, 3).reshape(B, T, H * D)
        return self.out(o)


class ConvModule(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d, K = cfg["d_model"], cfg["conv_kernel_size"]
        self.norm = _Norm(d, "layer_norm", cfg["norm_eps"])
        self.pw1 = _Lin(d, 2 * d)
        self.dw = _DW(d, K)
        self.cnorm = _Norm(d, cfg["conv_norm"], cfg["norm_eps"])
        self.pw2 = _Lin(d, d)

    def forward(self, x):
        d, K = self.dw.weight.shape
        g = F.glu(self.pw1(self.norm(x)), dim=-1)
        c = F.conv1d(g.transpose(1, 2), self.dw.weight.unsqueeze(1), self.dw.bias, padding=(K - 1) // 2, groups=d)
        return self.pw2(F.silu(self.cnorm(c.transpose(1, 2))))


class Block(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.ff1, self.attn, self.conv, self.ff2 = FeedForward(cfg), Attention(cfg), ConvModule(cfg), FeedForward(cfg)
        self.norm_out = _Norm(cfg["d_model"], "layer_norm", cfg["norm_eps"])

    def forward(self, x):
        x = x + 0.5 * self.ff1(x)
        x = x + self.attn(x)
        x = x + self.conv(x)
        x = x + 0.5 * self.ff2(x)
        return self.norm_out(x)


class Decoder(nn.Module):
    def __init__(self, cfg, num_classes):
        super().__init__()
        d = cfg["d_model"]
        self.num_classes = num_classes
        self.norm = _Norm(d, "layer_norm", cfg["norm_eps"])
        self.ff = _Lin(d, num_classes)
        self.reproj = _Lin(num_classes, d) if cfg["self_conditioning"] else None

    def logits(self, x):
        return self.ff(self.norm(x))


class SCConformerXLRef(nn.Module):
    """`vocab_size` excludes the CTC blank; `decoder.num_classes = vocab_size + 1`, blank = last class."""

    def __init__(self, config=None, vocab_size=128, seed=0, blank_bias=0.0):
        super().__init__()
        cfg = make_config(**(config or {}))
        self.config = cfg
        self.subsampling = Subsampling(cfg)
        self.layers = nn.ModuleList([Block(cfg) for _ in range(cfg["n_layers"])])
        self.decoder = Decoder(cfg, vocab_size + 1)
        self.device = torch.device("cpu")
        init_parameters(self, seed, blank_bias)

    def forward(self, audio_signal):
        x = self.subsampling(audio_signal)
        n = len(self.layers)
        for i, blk in enumerate(self.layers):
            x = blk(x)
            if self.decoder.reproj is not None and i != n - 1:
                x = x + self.decoder.reproj(torch.softmax(self.decoder.logits(x), -1))
        return {"final_posteriors": F.log_softmax(self.decoder.logits(x), -1), "hidden": x}

    def print_total_params(self):
        print(f"Total params: {sum(p.numel() for p in self.parameters()) / 1e6:.2f}M")


def init_parameters(model, seed=0, blank_bias=0.0):
    """Seeded synthetic weights (no checkpoint exists offline): U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for matrices and
    conv kernels, small uniform biases, norm gains 1 + small noise; one generator walked in named_parameters order so
    the oracle and the HIP model (which copies this state_dict) agree bit for bit."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("norm.weight") or name.endswith("norm_out.weight") or name.endswith("cnorm.weight"):
                p.copy_(1.0 + 0.1 * (torch.rand(p.shape, generator=g) - 0.5))
            elif p.dim() == 1:
                p.copy_(0.1 * (torch.rand(p.shape, generator=g) - 0.5))
            else:
                fan_in = p[0].numel()
                bound = 1.0 / math.sqrt(fan_in)
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
        if blank_bias:
            model.decoder.ff.bias[-1] += blank_bias
