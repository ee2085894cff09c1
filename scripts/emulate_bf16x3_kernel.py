"""Lane-level CPU emulation of csrc/gemm_bf16x3.hip (EXPERIMENTAL kernel, written without a GPU run): the same index arithmetic — staging split
into three bf16 planes, the per-lane fragment addresses, the six MFMAs per tile pair, the C/D register map of the epilogue — with
v_mfma_f32_32x32x16_bf16 modelled from the operand / result layouts of /opt/skills/guides/cdna_hip_programming.md (A: lane (r = l & 31, h = l >> 5)
holds A[row r][k = 8h + j]; B: B[k = 8h + j][col r]; D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)).  Checks the result against
float64 X W^T (+ bias) on ragged sizes.  What it proves: the kernel's indexing is consistent with the documented layouts; what it cannot prove:
that the hardware agrees — that is tests/test_zz_experimental_bf16x3_gpu.py's job (it did: profiles/r04_bf16x3_kernel_first_run.log).
    python scripts/emulate_bf16x3_kernel.py"""
import numpy as np

BM = BN = 128
BK = 32


def bf16_bits(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def bf16_value(b):
    return (np.asarray(b).astype(np.uint32) << 16).view(np.float32)


def split3(x):
    t0 = bf16_bits(x)
    r1 = (x - bf16_value(t0)).astype(np.float32)
    t1 = bf16_bits(r1)
    r2 = (r1 - bf16_value(t1)).astype(np.float32)
    return t0, t1, bf16_bits(r2)


def stage_split(src, rows, row0, k0):
    """[3][128][BK] planes of the tile (zero rows past the matrix), indexed as the kernel's float4 loop does."""
    dst = np.zeros((3, BM, BK), np.uint16)
    for idx in range(BM * BK // 4):
        row, c4 = idx >> 3, (idx & 7) * 4
        v = src[row0 + row, k0 + c4:k0 + c4 + 4] if row0 + row < rows else np.zeros(4, np.float32)
        t = split3(v)
        for p in range(3):
            dst[p, row, c4:c4 + 4] = t[p]
    return dst


def stage_split_t(src, rows, row0, k0):
    """The !KCONT staging of the prefetching form: the source is stored [K][128-dimension]; float4 along the 128-dimension (guarded element-wise at
    the ragged edge), planes written transposed [128-dimension][K]."""
    dst = np.zeros((3, BM, BK), np.uint16)
    for idx in range(BM * BK // 4):
        krow, c4 = idx >> 5, (idx & 31) * 4
        v = np.zeros(4, np.float32)
        for j in range(4):
            if row0 + c4 + j < rows:
                v[j] = src[k0 + krow, row0 + c4 + j]
        t = split3(v)
        for p in range(3):
            for j in range(4):
                dst[p, c4 + j, krow] = t[p][j]
    return dst


def split_planes(src):
    """dyn_bf16x3_split: [3][rows][K] planes of src [rows][K]."""
    t = split3(src.reshape(-1))
    return np.stack([x.reshape(src.shape) for x in t])


def stage_planes(planes, rows, row0, k0):
    """The pre-split B stage (load_planes / store_planes): 1536 16-byte pieces per tile, plane-major."""
    dst = np.zeros((3, BM, BK), np.uint16)
    for idx in range(3 * BM * BK * 2 // 16):
        p, rem = idx >> 9, idx & 511
        row, c8 = rem >> 2, (rem & 3) * 8
        if row0 + row < rows:
            dst[p, row, c8:c8 + 8] = planes[p, row0 + row, k0 + c8:k0 + c8 + 8]
    return dst


def mfma_32x32x16(a_lanes, b_lanes, acc_lanes):
    """a_lanes, b_lanes: [64][8] bf16 values as float32; acc_lanes: [64][16] float32 (updated in place, float32 accumulation)."""
    A = np.zeros((32, 16), np.float64)
    B = np.zeros((16, 32), np.float64)
    for lane in range(64):
        r, h = lane & 31, lane >> 5
        A[r, 8 * h:8 * h + 8] = a_lanes[lane]
        B[8 * h:8 * h + 8, r] = b_lanes[lane]
    D = A @ B          # every product exact; the k sum in float64 here (the hardware's internal order is its own)
    for lane in range(64):
        r, h = lane & 31, lane >> 5
        for q in range(16):
            row = (q & 3) + 8 * (q >> 2) + 4 * h
            acc_lanes[lane, q] = np.float32(np.float64(acc_lanes[lane, q]) + D[row, r])


def kernel(X, W, bias, M, N, K, ta=False, tb=True, bpre=False):
    """X: A as stored ([M][K], or [K][M] when ta); W: B as stored ([N][K] when tb, else [K][N]); bpre: W is the pre-split planes [3][N][K]."""
    C = np.full((M, N), np.nan, np.float32)
    for by in range(-(-M // BM)):
        for bx in range(-(-N // BN)):
            bm, bn = by * BM, bx * BN
            acc = np.zeros((4, 2, 2, 64, 16), np.float32)          # [wave][mi][ni][lane][reg]
            for k0 in range(0, K, BK):
                sX = stage_split_t(X, M, bm, k0) if ta else stage_split(X, M, bm, k0)
                sW = stage_planes(W, N, bn, k0) if bpre else (stage_split(W, N, bn, k0) if tb else stage_split_t(W, N, bn, k0))
                for wave in range(4):
                    wm, wn = (wave >> 1) * 64, (wave & 1) * 64
                    for ks in range(BK // 16):
                        a = np.zeros((3, 2, 64, 8), np.float32)
                        b = np.zeros((3, 2, 64, 8), np.float32)
                        for lane in range(64):
                            r, h = lane & 31, lane >> 5
                            kk = ks * 16 + 8 * h
                            for p in range(3):
                                for t in range(2):
                                    a[p, t, lane] = bf16_value(sX[p, wm + t * 32 + r, kk:kk + 8])
                                    b[p, t, lane] = bf16_value(sW[p, wn + t * 32 + r, kk:kk + 8])
                        for mi in range(2):
                            for ni in range(2):
                                for i, j in ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)):
                                    mfma_32x32x16(a[i, mi], b[j, ni], acc[wave, mi, ni])
            for wave in range(4):
                wm, wn = (wave >> 1) * 64, (wave & 1) * 64
                for mi in range(2):
                    for ni in range(2):
                        for lane in range(64):
                            r, h = lane & 31, lane >> 5
                            col = bn + wn + ni * 32 + r
                            if col >= N:
                                continue
                            for q in range(16):
                                row = bm + wm + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h
                                if row < M:
                                    C[row, col] = acc[wave, mi, ni, lane, q] + (bias[col] if bias is not None else np.float32(0))
    return C


def main():
    rng = np.random.default_rng(3)
    for M, N, K, with_bias in ((128, 128, 64, False), (150, 200, 96, True), (33, 129, 32, True)):
        X = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32) if with_bias else None
        got = kernel(X, W, bias, M, N, K)
        ref = X.astype(np.float64) @ W.astype(np.float64).T + (bias.astype(np.float64) if with_bias else 0.0)
        assert not np.isnan(got).any(), "an output element was never written"
        err = np.abs(got - ref).max() / np.abs(ref).max()
        print(f"M={M} N={N} K={K} bias={with_bias}: max |err| / max |C| = {err:.2e}", flush=True)
        assert err < 2e-6, err
    # the other transpose forms of the prefetching kernel (dyn_gemm_bf16x3): operands stored transposed, staged through stage_split_t
    for M, N, K, ta, tb in ((150, 200, 64, False, False), (130, 131, 32, True, False), (33, 129, 64, True, True)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = (rng.standard_normal((K, N)) * 0.1).astype(np.float32)
        got = kernel(np.ascontiguousarray(A.T) if ta else A, np.ascontiguousarray(B.T) if tb else B, None, M, N, K, ta, tb)
        ref = A.astype(np.float64) @ B.astype(np.float64)
        assert not np.isnan(got).any()
        err = np.abs(got - ref).max() / np.abs(ref).max()
        print(f"M={M} N={N} K={K} trans_a={ta} trans_b={tb}: max |err| / max |C| = {err:.2e}", flush=True)
        assert err < 2e-6, err
    # pre-split weight planes (dyn_bf16x3_split + dyn_gemm_bf16x3_presplit)
    for M, N, K in ((150, 200, 64), (33, 129, 32)):
        X = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
        got = kernel(X, split_planes(W), None, M, N, K, False, True, True)
        ref = X.astype(np.float64) @ W.astype(np.float64).T
        assert not np.isnan(got).any()
        err = np.abs(got - ref).max() / np.abs(ref).max()
        print(f"M={M} N={N} K={K} pre-split W planes: max |err| / max |C| = {err:.2e}", flush=True)
        assert err < 2e-6, err
    print("emulation matches float64 within fp32 rounding: the kernel's indexing is consistent with the documented MFMA layouts")


if __name__ == "__main__":
    main()
