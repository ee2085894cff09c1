// EXPERIMENTAL — not on the product path: nothing but tests/test_zz_experimental_bf16x3_gpu.py and scripts/probe_gemm_bf16x3.py calls it (DESIGN.md section 6,
// "where the next factor is").
//
// An fp32 GEMM on the bf16 matrix cores by OPERAND SPLITTING: every fp32 operand is exactly the sum of three bf16 terms (3 x 8 significant
// bits), a product keeps the six terms a_i * b_j with i + j <= 2 (the dropped ones are below 2^-24 relative), each bf16 x bf16 product is
// exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, so the only roundings are fp32 accumulations: fp32-grade results
// (scripts/probe_bf16x3_numerics.py, tests/bf16x3_parity_study.py) at six dense bf16 MFMAs per fp32 product — a ceiling of ~417 TFLOP/s
// fp32-equivalent against 157 TFLOP/s for v_mfma_f32_32x32x2_f32.
//
// This file is the first, correctness-first form of that kernel, written at the end of round 4: its index arithmetic was checked by a lane-level
// emulation on the CPU (scripts/emulate_bf16x3_kernel.py), then it ran on the MI355X with the round's last GPU seconds
// (tests/test_zz_experimental_bf16x3_gpu.py: five shapes, each closer to float64 than dyn_gemm_f32; scripts/probe_gemm_bf16x3.py: 83 - 108 TFLOP/s
// fp32-equivalent at M >= 8192 against 120 - 131 for the tuned fp32 kernel — profiles/r04_bf16x3_kernel_first_*.log).
// Shape: C[M, N] = X[M, K] . W[N, K]^T (+ bias[N]) — the `linear` layout of the path (reference: every nn.Linear inside
// model(audio_signal=...), lcasr/lib.py:550), both operands K-contiguous, which is also what the MFMA fragment wants: lane (r = l & 31,
// h = l >> 5) holds X[row r][k = 8h + j] and W[col r][k = 8h + j], j = 0..7 — 8 consecutive k of one row.
//
// 128 x 128 tile per 256-thread workgroup, one 64 x 64 quadrant (2 x 2 MFMA tiles, 64 accumulator VGPRs) per wave, BK = 32 per LDS stage.
// The split happens ONCE per staged element on the way into LDS (three bf16 planes per operand, row stride padded by 16 B).  No software
// pipelining, no direct-to-LDS loads, no XCD mapping yet: those are the next steps once the numbers of this form are known.
#include <cstdlib>
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128, BN = 128, BK = 32, NTHREADS = 256;
constexpr int LDK = BK + 8;   // row stride in bf16 elements: 80 B = 5 x 16 B (fragments stay 16-B aligned, rows spread over the banks)

// fp32 -> nearest-even bf16 bits.  Integer form: a NaN input may come out as 0 / inf (MI355X_MICROARCH.md); operands here are finite activations
// and weights — the production form should use the cvt instruction.
__device__ __forceinline__ unsigned short bf16_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_value(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// x = t0 + t1 + t2 exactly for |x| >= 2^-95 (each residual is exact in fp32: the subtrahend is x's leading bits; below ~2^-102 the third term
// would be a bf16 subnormal and the sum is off by < 2^-133 absolute — tests/test_host_cpu.py::test_bf16x3_split_is_exact_on_the_host_emulation)
__device__ __forceinline__ void split3(float x, unsigned short& t0, unsigned short& t1, unsigned short& t2) {
    t0 = bf16_bits(x);
    const float r1 = x - bf16_value(t0);
    t1 = bf16_bits(r1);
    const float r2 = r1 - bf16_value(t1);
    t2 = bf16_bits(r2);
}

// stage a [128][BK] fp32 tile (rows row0 .. row0 + 127 of a [rows][ld] matrix, columns k0 .. k0 + BK - 1) as three bf16 planes
__device__ __forceinline__ void stage_split(const float* __restrict__ src, int64_t ld, int64_t rows, int64_t row0, int64_t k0,
                                            unsigned short (*dst)[BM][LDK]) {
#pragma unroll
    for (int i = 0; i < (BM * BK / 4) / NTHREADS; ++i) {   // 1024 float4 per tile, 4 per thread
        const int idx = threadIdx.x + i * NTHREADS;
        const int row = idx >> 3, c4 = (idx & 7) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < rows) v = *reinterpret_cast<const float4*>(src + (row0 + row) * ld + k0 + c4);
        const float e[4] = {v.x, v.y, v.z, v.w};
        unsigned short t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split3(e[j], t[0][j], t[1][j], t[2][j]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            uint2 w;
            w.x = (unsigned)t[p][0] | ((unsigned)t[p][1] << 16);
            w.y = (unsigned)t[p][2] | ((unsigned)t[p][3] << 16);
            *reinterpret_cast<uint2*>(&dst[p][row][c4]) = w;   // 8-B aligned: c4 * 2 B is a multiple of 8, the row stride of 16
        }
    }
}

// ---- variant 2 (NOT YET RUN ON HARDWARE — selected only by DYN_BF16X3_VARIANT=2): the same tile and fragment layout, but (a) the global loads of
// tile k + 1 are issued before the MFMAs of tile k and land in registers while they run (the first form waits for every load), and (b) the split
// uses the hardware conversion (a plain cast compiles to v_cvt_pk_bf16_f32 at -O3: round-to-nearest-even like bf16_bits on finite values, ~1/3
// of the integer form's VALU work).
__device__ __forceinline__ void split3_cvt(float x, unsigned short& t0, unsigned short& t1, unsigned short& t2) {
    const __bf16 h0 = (__bf16)x;
    const float r1 = x - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const float r2 = r1 - (float)h1;
    const __bf16 h2 = (__bf16)r2;
    t0 = __builtin_bit_cast(unsigned short, h0);
    t1 = __builtin_bit_cast(unsigned short, h1);
    t2 = __builtin_bit_cast(unsigned short, h2);
}

constexpr int F4_PER_THREAD = (BM * BK / 4) / NTHREADS;   // 4

// KCONT: the tile's 128-dimension is the ROW index of the source and K is contiguous (X of X W^T, W stored [N][K]): float4 along K.
// !KCONT: the source is stored [K][128-dimension] (a transposed A, or B stored [K][N]): float4 along the 128-dimension, guarded element-wise at
// the ragged edge; the planes are still written [128-dimension][K], so the fragment reads and the MFMA loop do not change.
template <bool KCONT>
__device__ __forceinline__ void load_tile(const float* __restrict__ src, int64_t ld, int64_t rows, int64_t row0, int64_t k0, float4 (&v)[F4_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < F4_PER_THREAD; ++i) {
        const int idx = threadIdx.x + i * NTHREADS;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KCONT) {
            const int row = idx >> 3, c4 = (idx & 7) * 4;
            if (row0 + row < rows) v[i] = *reinterpret_cast<const float4*>(src + (row0 + row) * ld + k0 + c4);
        } else {
            const int krow = idx >> 5, c4 = (idx & 31) * 4;
            const float* q = src + (k0 + krow) * ld + row0 + c4;
            if (row0 + c4 + 3 < rows) v[i] = *reinterpret_cast<const float4*>(q);
            else {
                if (row0 + c4 + 0 < rows) v[i].x = q[0];
                if (row0 + c4 + 1 < rows) v[i].y = q[1];
                if (row0 + c4 + 2 < rows) v[i].z = q[2];
            }
        }
    }
}

template <bool KCONT>
__device__ __forceinline__ void store_split(const float4 (&v)[F4_PER_THREAD], unsigned short (*dst)[BM][LDK]) {
#pragma unroll
    for (int i = 0; i < F4_PER_THREAD; ++i) {
        const int idx = threadIdx.x + i * NTHREADS;
        const float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        unsigned short t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split3_cvt(e[j], t[0][j], t[1][j], t[2][j]);
        if (KCONT) {
            const int row = idx >> 3, c4 = (idx & 7) * 4;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                uint2 w;
                w.x = (unsigned)t[p][0] | ((unsigned)t[p][1] << 16);
                w.y = (unsigned)t[p][2] | ((unsigned)t[p][3] << 16);
                *reinterpret_cast<uint2*>(&dst[p][row][c4]) = w;
            }
        } else {
            const int krow = idx >> 5, c4 = (idx & 31) * 4;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[p][c4 + j][krow] = t[p][j];   // transposing 2-byte stores (first form: bank conflicts not yet looked at)
        }
    }
}

// ---- pre-split B operand (NOT YET RUN ON HARDWARE): the weights change once per optimiser step, so their three bf16 planes can be written once
// per step (dyn_bf16x3_split; or by the optimiser kernel itself) instead of being split by every workgroup that stages them — that halves the
// split work on the critical path and makes the B stage a plain 16-byte copy.  Planes: unsigned short [3][rows][K], K contiguous.
constexpr int U4_PER_THREAD = (3 * BN * BK * 2 / 16) / NTHREADS;   // 1536 16-byte pieces per tile, 6 per thread

__device__ __forceinline__ void load_planes(const unsigned short* __restrict__ planes, int64_t rows, int64_t K, int64_t row0, int64_t k0,
                                            uint4 (&v)[U4_PER_THREAD]) {
#pragma unroll
    for (int i = 0; i < U4_PER_THREAD; ++i) {
        const int idx = threadIdx.x + i * NTHREADS;          // 0 .. 1535
        const int p = idx >> 9, rem = idx & 511;             // 512 pieces per plane
        const int row = rem >> 2, c8 = (rem & 3) * 8;
        v[i] = make_uint4(0u, 0u, 0u, 0u);
        if (row0 + row < rows) v[i] = *reinterpret_cast<const uint4*>(planes + ((int64_t)p * rows + row0 + row) * K + k0 + c8);
    }
}

__device__ __forceinline__ void store_planes(const uint4 (&v)[U4_PER_THREAD], unsigned short (*dst)[BN][LDK]) {
#pragma unroll
    for (int i = 0; i < U4_PER_THREAD; ++i) {
        const int idx = threadIdx.x + i * NTHREADS;
        const int p = idx >> 9, rem = idx & 511;
        const int row = rem >> 2, c8 = (rem & 3) * 8;
        *reinterpret_cast<uint4*>(&dst[p][row][c8]) = v[i];   // 16-B aligned: row stride 80 B, c8 * 2 B in {0, 16, 32, 48}
    }
}

__global__ __launch_bounds__(NTHREADS) void bf16x3_split_kernel(const float* __restrict__ src, unsigned short* __restrict__ planes, int64_t rows,
                                                                int64_t K, int64_t ld) {
    const int64_t n4 = rows * (K / 4);
    for (int64_t i = (int64_t)blockIdx.x * NTHREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * NTHREADS) {
        const int64_t row = i / (K / 4), c4 = (i % (K / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + row * ld + c4);
        const float e[4] = {v.x, v.y, v.z, v.w};
        unsigned short t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split3_cvt(e[j], t[0][j], t[1][j], t[2][j]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            uint2 w;
            w.x = (unsigned)t[p][0] | ((unsigned)t[p][1] << 16);
            w.y = (unsigned)t[p][2] | ((unsigned)t[p][3] << 16);
            *reinterpret_cast<uint2*>(planes + ((int64_t)p * rows + row) * K + c4) = w;
        }
    }
}

// TA: A stored [K][M]; TB: B stored [N][K] (the flags of dyn_gemm_f32).  The first, non-prefetching form exists for X W^T only (!TA, TB).
// BPRE: B comes as pre-split planes [3][N][K] (W is then that pointer; needs PIPE, !TA-or-TA as usual, TB).
template <bool PIPE, bool TA, bool TB, bool BPRE = false>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16x3_nt_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                                  const float* __restrict__ bias, float* __restrict__ C, int64_t M,
                                                                  int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldc) {
    __shared__ __attribute__((aligned(16))) unsigned short sX[3][BM][LDK];   // 30 720 B
    __shared__ __attribute__((aligned(16))) unsigned short sW[3][BN][LDK];   // 30 720 B
    const int64_t bm = (int64_t)blockIdx.y * BM, bn = (int64_t)blockIdx.x * BN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;

    float4 px[F4_PER_THREAD], pw[F4_PER_THREAD];
    uint4 pp[U4_PER_THREAD];
    static_assert(PIPE || (!TA && TB), "the non-prefetching form stages K-contiguous operands only");
    static_assert(!BPRE || (PIPE && TB), "pre-split B planes are [N][K] and go through the prefetching form");
    const unsigned short* Wp = reinterpret_cast<const unsigned short*>(W);
    if (PIPE) {
        load_tile<!TA>(X, ldx, M, bm, 0, px);
        if (BPRE) load_planes(Wp, N, K, bn, 0, pp);
        else load_tile<TB>(W, ldw, N, bn, 0, pw);
    }
    for (int64_t k0 = 0; k0 < K; k0 += BK) {
        if (PIPE) {
            store_split<!TA>(px, sX);
            if (BPRE) store_planes(pp, sW);
            else store_split<TB>(pw, sW);
            __syncthreads();
            if (k0 + BK < K) {   // in flight during the MFMAs below
                load_tile<!TA>(X, ldx, M, bm, k0 + BK, px);
                if (BPRE) load_planes(Wp, N, K, bn, k0 + BK, pp);
                else load_tile<TB>(W, ldw, N, bn, k0 + BK, pw);
            }
        } else {
            stage_split(X, ldx, M, bm, k0, sX);
            stage_split(W, ldw, N, bn, k0, sW);
            __syncthreads();
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int kk = ks * 16 + 8 * h;
            bf16x8 a[3][2], b[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[p][t] = *reinterpret_cast<const bf16x8*>(&sX[p][wm + t * 32 + r][kk]);
                    b[p][t] = *reinterpret_cast<const bf16x8*>(&sW[p][wn + t * 32 + r][kk]);
                }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    f32x16 c = acc[mi][ni];
                    // the three smallest terms first, the leading product last (the order of the CPU emulations)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[2][ni], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][mi], b[0][ni], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mi], b[1][ni], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[1][ni], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][mi], b[0][ni], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[0][ni], c, 0, 0, 0);
                    acc[mi][ni] = c;
                }
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int64_t col = bn + wn + ni * 32 + r;
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t row = bm + wm + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (row < M) C[row * ldc + col] = acc[mi][ni][q] + bv;
            }
        }
}

}  // namespace

// C[M, N] = op(A) . op(B) (+ bias[N]) in fp32-grade arithmetic on the bf16 matrix cores (see the header of this file: EXPERIMENTAL).
// trans_a: A stored [K][M]; trans_b: B stored [N][K] (dyn_gemm_f32's flags: the linear layer's forward is (0, 1), its input gradient (0, 0),
// its weight gradient (1, 0)).  K must be a multiple of 32; A, B 16-byte aligned with lda, ldb multiples of 4.  Operands must be finite.
// Only (0, 1) in its first form (DYN_BF16X3_VARIANT unset) has run on hardware; every other combination goes through the prefetching form.
extern "C" int dyn_gemm_bf16x3(int trans_a, int trans_b, const float* A, const float* B, const float* bias, float* C, int64_t M, int64_t N,
                               int64_t K, int64_t lda, int64_t ldb, int64_t ldc, void* stream) {
    DYN_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, DYN_E_ARG, "dyn_gemm_bf16x3: bad arguments");
    DYN_REQUIRE(K % BK == 0, DYN_E_UNSUPPORTED, "dyn_gemm_bf16x3: K = %lld is not a multiple of %d", (long long)K, BK);
    DYN_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N && lda % 4 == 0 && ldb % 4 == 0 && (((uintptr_t)A) & 15) == 0 &&
                    (((uintptr_t)B) & 15) == 0,
                DYN_E_ARG, "dyn_gemm_bf16x3: operands must be 16-byte aligned with leading dimensions that are multiples of 4 and >= the row length");
    const int64_t gx = dyn::cdiv(N, BN), gy = dyn::cdiv(M, BM);
    DYN_REQUIRE(gx < 65536 * 16 && gy < 65536, DYN_E_ARG, "dyn_gemm_bf16x3: grid too large");
    static const int variant = [] { const char* e = getenv("DYN_BF16X3_VARIANT"); return e ? atoi(e) : 1; }();
    const dim3 grid((unsigned)gx, (unsigned)gy), blk(NTHREADS);
    hipStream_t st = (hipStream_t)stream;
#define GO(P, TA_, TB_) hipLaunchKernelGGL((gemm_bf16x3_nt_kernel<P, TA_, TB_>), grid, blk, 0, st, A, B, bias, C, M, N, K, lda, ldb, ldc)
    if (!trans_a && trans_b) {
        if (variant == 2) GO(true, false, true);   // prefetching form: written after the round's last GPU run, so it has not executed yet
        else GO(false, false, true);
    } else if (!trans_a && !trans_b) GO(true, false, false);
    else if (trans_a && !trans_b) GO(true, true, false);
    else GO(true, true, true);
#undef GO
    return dyn::check_launch("dyn_gemm_bf16x3");
}

extern "C" int dyn_gemm_bf16x3_nt(const float* X, const float* W, const float* bias, float* C, int64_t M, int64_t N, int64_t K, int64_t ldx,
                                  int64_t ldw, int64_t ldc, void* stream) {
    return dyn_gemm_bf16x3(0, 1, X, W, bias, C, M, N, K, ldx, ldw, ldc, stream);
}

// planes [3][rows][K] (unsigned short, K contiguous) <- the three bf16 terms of src [rows][K] (ld >= K, multiple of 4; K % 4 == 0).  NOT YET RUN.
extern "C" int dyn_bf16x3_split(const float* src, void* planes, int64_t rows, int64_t K, int64_t ld, void* stream) {
    DYN_REQUIRE(src && planes && rows > 0 && K > 0 && K % 4 == 0 && ld >= K && ld % 4 == 0 && (((uintptr_t)src) & 15) == 0 &&
                    (((uintptr_t)planes) & 15) == 0,
                DYN_E_ARG, "dyn_bf16x3_split: bad arguments");
    int64_t g = dyn::cdiv(rows * (K / 4), NTHREADS);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(bf16x3_split_kernel, dim3((unsigned)g), dim3(NTHREADS), 0, (hipStream_t)stream, src, (unsigned short*)planes, rows, K, ld);
    return dyn::check_launch("dyn_bf16x3_split");
}

// C[M, N] = X[M, K] . W^T (+ bias) with W given as the pre-split planes of dyn_bf16x3_split (rows = N).  K % 32 == 0.  NOT YET RUN.
extern "C" int dyn_gemm_bf16x3_presplit(const float* X, const void* w_planes, const float* bias, float* C, int64_t M, int64_t N, int64_t K,
                                        int64_t ldx, int64_t ldc, void* stream) {
    DYN_REQUIRE(X && w_planes && C && M > 0 && N > 0 && K > 0, DYN_E_ARG, "dyn_gemm_bf16x3_presplit: bad arguments");
    DYN_REQUIRE(K % BK == 0, DYN_E_UNSUPPORTED, "dyn_gemm_bf16x3_presplit: K = %lld is not a multiple of %d", (long long)K, BK);
    DYN_REQUIRE(ldx >= K && ldc >= N && ldx % 4 == 0 && (((uintptr_t)X) & 15) == 0 && (((uintptr_t)w_planes) & 15) == 0, DYN_E_ARG,
                "dyn_gemm_bf16x3_presplit: operands must be 16-byte aligned, ldx a multiple of 4");
    const int64_t gx = dyn::cdiv(N, BN), gy = dyn::cdiv(M, BM);
    DYN_REQUIRE(gx < 65536 * 16 && gy < 65536, DYN_E_ARG, "dyn_gemm_bf16x3_presplit: grid too large");
    hipLaunchKernelGGL((gemm_bf16x3_nt_kernel<true, false, true, true>), dim3((unsigned)gx, (unsigned)gy), dim3(NTHREADS), 0, (hipStream_t)stream, X,
                       (const float*)w_planes, bias, C, M, N, K, ldx, K, ldc);
    return dyn::check_launch("dyn_gemm_bf16x3_presplit");
}
