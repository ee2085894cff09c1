"""Fused (streaming) vs unfused (materialised scores) attention, forward + backward, timed with HIP events on one GPU.
usage: python scripts/probe_attention.py [T ...]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd import ops

dev = torch.device("cuda:0")
H, D = 6, 128
HD = H * D


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for T in [int(a) for a in sys.argv[1:]] or [2048, 8192]:
    scale = 1.0 / math.sqrt(D)
    B_f, B_b = 2, 1
    qkv = torch.randn(B_f, T, 3 * HD, device=dev)
    dO = torch.randn(B_b, T, HD, device=dev)

    def unf_fwd():
        S = torch.empty(B_f, H, T, T, device=dev)
        ops.gemm(qkv, qkv, S, trans_b=True, M=T, N=T, K=D, lda=3 * HD, ldb=3 * HD, ldc=T, nb1=B_f, nb2=H, sa=(T * 3 * HD, D), sb=(T * 3 * HD, D),
                 sc=(H * T * T, T * T), b_off=HD, alpha=scale)
        ops.softmax(S, out=S)
        O = torch.empty(B_f, T, HD, device=dev)
        ops.gemm(S, qkv, O, M=T, N=D, K=T, lda=T, ldb=3 * HD, ldc=HD, nb1=B_f, nb2=H, sa=(H * T * T, T * T), sb=(T * 3 * HD, D), sc=(T * HD, D), b_off=2 * HD)
        return S, O

    S, O = unf_fwd()
    q1 = qkv[:B_b].contiguous(); S1 = S[:B_b].contiguous()

    def unf_bwd():
        dqkv = torch.empty_like(q1)
        sS, sQ, sO = (H * T * T, T * T), (T * 3 * HD, D), (T * HD, D)
        ops.gemm(S1, dO, dqkv, trans_a=True, M=T, N=D, K=T, lda=T, ldb=HD, ldc=3 * HD, nb1=B_b, nb2=H, sa=sS, sb=sO, sc=sQ, c_off=2 * HD)
        dP = torch.empty_like(S1)
        ops.gemm(dO, q1, dP, trans_b=True, M=T, N=T, K=D, lda=HD, ldb=3 * HD, ldc=T, nb1=B_b, nb2=H, sa=sO, sb=sQ, sc=sS, b_off=2 * HD)
        ops.softmax_bwd(S1, dP, out=dP, scale=1.0)
        ops.gemm(dP, q1, dqkv, M=T, N=D, K=T, lda=T, ldb=3 * HD, ldc=3 * HD, nb1=B_b, nb2=H, sa=sS, sb=sQ, sc=sQ, b_off=HD, c_off=0, alpha=scale)
        ops.gemm(dP, q1, dqkv, trans_a=True, M=T, N=D, K=T, lda=T, ldb=3 * HD, ldc=3 * HD, nb1=B_b, nb2=H, sa=sS, sb=sQ, sc=sQ, b_off=0, c_off=HD, alpha=scale)

    O2, lse = ops.attention_fwd(qkv, B_f, T, H, D, scale, want_lse=True)
    O1, lse1 = O2[:B_b].contiguous(), lse[:B_b].contiguous()
    t_uf, t_ub = timeit(unf_fwd), timeit(unf_bwd)
    t_ff = timeit(lambda: ops.attention_fwd(qkv, B_f, T, H, D, scale, want_lse=True))
    t_fb = timeit(lambda: ops.attention_bwd(q1, O1, dO, lse1, B_b, T, H, D, scale))
    fl_f = 4.0 * B_f * H * T * T * D
    fl_b4, fl_b7 = 8.0 * B_b * H * T * T * D, 14.0 * B_b * H * T * T * D
    print(f"T'={T}: forward B={B_f}: unfused {t_uf*1e3:.0f} us ({fl_f/t_uf/1e9:.0f} TF/s) | fused {t_ff*1e3:.0f} us ({fl_f/t_ff/1e9:.0f} TF/s)   "
          f"backward B={B_b}: unfused {t_ub*1e3:.0f} us ({fl_b4/t_ub/1e9:.0f} TF/s on 4 products) | fused {t_fb*1e3:.0f} us "
          f"({fl_b7/t_fb/1e9:.0f} TF/s on 7 products, {fl_b4/t_fb/1e9:.0f} useful)   scores kept: unfused {B_f*H*T*T*4/1e9:.2f} GB, fused {B_f*H*T*4/1e6:.2f} MB", flush=True)
    del S, S1

# no-grad forward with key splits (dyn_attention_fwd_split): the final pass (B = 4), the adapt step (B = 2), plain inference (B = 1)
for T in [int(a) for a in sys.argv[1:]] or [2048]:
    scale = 1.0 / math.sqrt(D)
    for B in (4, 2, 1):
        qkv = torch.randn(B, T, 3 * HD, device=dev)
        fl = 4.0 * B * H * T * T * D
        line = f"T'={T} no-grad forward B={B} ({B * H * ((T + 127) // 128)} query-block workgroups):"
        for ns in (1, 0, 2, 3, 4, 6, 8):
            try:
                t = timeit(lambda: ops.attention_fwd(qkv, B, T, H, D, scale, nsplit=ns), n=20)
            except Exception as e:
                line += f"  nsplit={ns}: n/a"
                continue
            line += f"  nsplit={'auto' if ns == 0 else ns}: {t*1e3:.0f} us ({fl/t/1e9:.0f} TF/s)"
        print(line, flush=True)
