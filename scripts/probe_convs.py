"""Times the subsampling kernels at the window shapes (B=2: x [2,16384,80] -> z1 [2,8192,40,256] -> u2 [2,4096,20,256])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_asr_eval_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (2, 1):
    C = 256
    x = torch.randn(B, 16384, 80, device=dev)
    w1 = torch.randn(C, 3, 3, device=dev); b1 = torch.randn(C, device=dev)
    z1 = ops.conv2d_first(x, w1, b1)
    u2 = ops.dwconv2d_s2(z1, w1, b1)
    du2 = torch.randn_like(u2); dz1 = torch.randn_like(z1)
    dw = torch.zeros(C, 3, 3, device=dev); db = torch.zeros(C, device=dev)
    GB = lambda *ts: sum(t.numel() for t in ts) * 4 / 1e9
    for name, fn, gb in (("conv2d_first fwd", lambda: ops.conv2d_first(x, w1, b1, out=z1), GB(x, z1)),
                         ("dwconv2d_s2 fwd", lambda: ops.dwconv2d_s2(z1, w1, b1, out=u2), GB(z1, u2)),
                         ("dwconv2d_s2 dgrad", lambda: ops.dwconv2d_s2_dgrad(z1, w1, du2, out=dz1), GB(z1, du2, dz1)),
                         ("dwconv2d_s2 wgrad", lambda: ops.dwconv2d_s2_wgrad(z1, du2, dw, db, beta=0.0), GB(z1, du2)),
                         ("conv2d_first wgrad", lambda: ops.conv2d_first_wgrad(x, dz1, dw, db, beta=0.0), GB(x, dz1))):
        us = timeit(fn)
        print(f"B={B} {name:20s} {us:8.1f} us  {gb/us*1e6/1e3:6.2f} TB/s algorithmic ({gb*1e3:.0f} MB)", flush=True)
    # the same two stages fused (z1 / dz1 recomputed, never stored): forward replaces the first two lines, backward the last three
    dw2 = torch.zeros(C, 3, 3, device=dev); db2 = torch.zeros(C, device=dev)
    us = timeit(lambda: ops.sub12_fwd(x, w1, b1, w1, b1, out=u2))
    print(f"B={B} {'sub12 fused fwd':20s} {us:8.1f} us  (reads {GB(x)*1e3:.0f} MB, writes {GB(u2)*1e3:.0f} MB)", flush=True)
    us = timeit(lambda: ops.sub12_bwd(x, du2, w1, b1, w1, dw, db, dw2, db2, beta=0.0))
    print(f"B={B} {'sub12 fused bwd':20s} {us:8.1f} us  (reads {GB(x, du2)*1e3:.0f} MB)", flush=True)
for rows, C in ((2048, 768), (2048, 3072), (2048, 4096), (8192, 768)):
    x = torch.randn(rows, C, device=dev); out = torch.zeros(C, device=dev)
    us = timeit(lambda: ops.colsum(x, out, beta=1.0), n=50)
    print(f"colsum [{rows},{C}] {us:7.1f} us  {rows*C*4/us/1e6:6.2f} TB/s", flush=True)
for B in (2, 1):
    C = 768
    u = torch.randn(B, 2048, 2 * C, device=dev); wc = torch.randn(C, 9, device=dev); bc = torch.randn(C, device=dev); gm = torch.ones(C, device=dev)
    for save in (True, False):
        us = timeit(lambda: ops.convmod_fwd(u, wc, bc, gm, None, False, 1e-5, save), n=30)
        mb = (u.numel() + B * 2048 * C * (4 if save else 1)) * 4 / 1e6
        print(f"convmod_fwd B={B} save={save}: {us:7.1f} us  {mb/us*1e6/1e6:6.2f} TB/s ({mb:.0f} MB)", flush=True)
import math
for B in (4, 2, 1):
    H, D, T = 6, 128, 2048
    qkv = torch.randn(B, T, 3 * H * D, device=dev)
    HD = H * D
    S = torch.empty(B, H, T, T, device=dev); O = torch.empty(B, T, HD, device=dev)
    def unfused():
        ops.gemm(qkv, qkv, S, trans_b=True, M=T, N=T, K=D, lda=3 * HD, ldb=3 * HD, ldc=T, nb1=B, nb2=H,
                 sa=(T * 3 * HD, D), sb=(T * 3 * HD, D), sc=(H * T * T, T * T), b_off=HD, alpha=1.0 / math.sqrt(D))
        ops.softmax(S, out=S)
        ops.gemm(S, qkv, O, M=T, N=D, K=T, lda=T, ldb=3 * HD, ldc=HD, nb1=B, nb2=H, sa=(H * T * T, T * T), sb=(T * 3 * HD, D), sc=(T * HD, D), b_off=2 * HD)
    tu = timeit(unfused, n=20)
    tf = timeit(lambda: ops.attention_fwd(qkv, B, T, H, D, 1.0 / math.sqrt(D), out=O), n=20)
    fl = 4.0 * B * H * T * T * D
    print(f"attention B={B}: unfused {tu:7.1f} us ({fl/tu/1e6:5.1f} TF/s)  fused {tf:7.1f} us ({fl/tf/1e6:5.1f} TF/s)", flush=True)
