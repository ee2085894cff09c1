"""EXPERIMENTAL (round 5 starts here): time dyn_gemm_bf16x3_nt against dyn_gemm_f32 on the path's linear-layer shapes and print the error of both
against float64.  python scripts/probe_gemm_bf16x3.py   (GPU box; DYN_BF16X3_VARIANT=2 selects the prefetching form, `all` as an argument also times
the input-gradient (0, 0) and weight-gradient (1, 0) forms, which have not run on hardware yet)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dynamic_asr_eval_amd import ops  # noqa: E402
from dynamic_asr_eval_amd._lib import check, load  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in ((4096, 768, 768), (4096, 3072, 768), (4096, 768, 3072), (4096, 2304, 768), (16384, 768, 768), (16384, 3072, 768), (8192, 4096, 768)):
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05
    c3, st = torch.empty(M, N, device=dev), torch.cuda.current_stream().cuda_stream
    t3 = timeit(lambda: check(load().dyn_gemm_bf16x3_nt(x.data_ptr(), w.data_ptr(), None, c3.data_ptr(), M, N, K, K, K, N, st), "bf16x3"))
    c1 = torch.empty(M, N, device=dev)
    t1 = timeit(lambda: ops.linear(x, w, None, out=c1))
    ref = x.double() @ w.double().t()
    s = ref.abs().max().item()
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}: bf16x3 {t3 * 1e3:.1f} us = {fl / t3 / 1e9:.1f} TFLOP/s fp32-equivalent (err {(c3.double() - ref).abs().max().item() / s:.1e}) | "
          f"dyn_gemm_f32 {t1 * 1e3:.1f} us = {fl / t1 / 1e9:.1f} TFLOP/s (err {(c1.double() - ref).abs().max().item() / s:.1e})", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "all":
    for M, N, K, ta, tb in ((4096, 768, 768, 0, 0), (4096, 768, 3072, 0, 0), (16384, 768, 3072, 0, 0), (768, 768, 4096, 1, 0), (3072, 768, 4096, 1, 0), (768, 3072, 16384, 1, 0)):
        a, b = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev) * 0.05
        A, B = (a.t().contiguous() if ta else a), (b.t().contiguous() if tb else b)
        c3, c1, st = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev), torch.cuda.current_stream().cuda_stream
        t3 = timeit(lambda: check(load().dyn_gemm_bf16x3(ta, tb, A.data_ptr(), B.data_ptr(), None, c3.data_ptr(), M, N, K, A.shape[1], B.shape[1], N, st), "bf16x3"))
        t1 = timeit(lambda: ops.gemm(A, B, c1, trans_a=bool(ta), trans_b=bool(tb), M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], ldc=N))
        ref = a.double() @ b.double()
        s = ref.abs().max().item()
        fl = 2.0 * M * N * K
        print(f"ta={ta} tb={tb} {M}x{N}x{K}: bf16x3 {t3 * 1e3:.1f} us = {fl / t3 / 1e9:.1f} TFLOP/s fp32-equivalent (err {(c3.double() - ref).abs().max().item() / s:.1e}) | "
              f"dyn_gemm_f32 {t1 * 1e3:.1f} us = {fl / t1 / 1e9:.1f} TFLOP/s (err {(c1.double() - ref).abs().max().item() / s:.1e})", flush=True)
    # pre-split weight planes: the B stage becomes a copy, only X is split by the workgroups
    for M, N, K in ((4096, 768, 768), (4096, 3072, 768), (16384, 768, 768), (16384, 3072, 768)):
        x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05
        planes, c3, st = torch.empty(3, N, K, dtype=torch.int16, device=dev), torch.empty(M, N, device=dev), torch.cuda.current_stream().cuda_stream
        ts = timeit(lambda: check(load().dyn_bf16x3_split(w.data_ptr(), planes.data_ptr(), N, K, K, st), "split"))
        t3 = timeit(lambda: check(load().dyn_gemm_bf16x3_presplit(x.data_ptr(), planes.data_ptr(), None, c3.data_ptr(), M, N, K, K, N, st), "presplit"))
        ref = x.double() @ w.double().t()
        fl = 2.0 * M * N * K
        print(f"pre-split {M}x{N}x{K}: split of W {ts * 1e3:.1f} us, product {t3 * 1e3:.1f} us = {fl / t3 / 1e9:.1f} TFLOP/s fp32-equivalent "
              f"(err {(c3.double() - ref).abs().max().item() / ref.abs().max().item():.1e})", flush=True)
