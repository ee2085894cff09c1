// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-exact fmaf chain).
//
// Replaces the dense products inside model(audio_signal=...) and loss.backward() of the reference's
// dynamic-eval loop (reference lcasr/lib.py:550,579,603): QKV/out projections, FFN, pointwise convs,
// subsampling projections, CTC head, self-conditioning re-projection, and every dgrad/wgrad of those.
//
// Design (MI355X-first, not a warp-tiled port):
//   * 256-thread workgroup = 4 wave64 as 2(M) x 2(N); each wave owns (BM/2)x(BN/2) as 32x32 MFMA tiles,
//     accumulators stay in registers for the whole K loop (64 VGPRs at 128x128).
//   * BK = 32.  An operand whose K is contiguous in HBM is staged as [row][BK+4]: one ds_read_b128 per lane
//     feeds FOUR k-steps (lane half h covers k = 8c+4h+s), and the 36-float row stride keeps every
//     16-lane ds_read_b128 group on 16 distinct 16-B slots (conflict-free, MI355X_MICROARCH.md §LDS).
//     An operand whose ROW index is contiguous in HBM (transposed use) is staged as [k][rows+4] and read with
//     conflict-free ds_read_b32 (32 consecutive floats per lane half) using the same k(c,h,s) order.
//   * global -> registers -> LDS double buffer: tile t+1's 16-B global loads are issued before tile t's MFMAs
//     and written to the other LDS buffer after them; one barrier per K tile.
//   * XCD-aware bijective block remap: the 8 XCDs (private L2s) each walk a contiguous run of output tiles.
//   * split-K writes fp32 slabs to a caller workspace and a second kernel sums them in slice order:
//     deterministic wgrad without float atomics.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;
constexpr int NTHREADS = 256;

struct KParams {
    int64_t M, N, K;
    const float* A; int64_t lda, sa1, sa2;
    const float* B; int64_t ldb, sb1, sb2;
    float* C; int64_t ldc, sc1, sc2;
    const float* bias;
    float alpha, beta;
    int64_t nb2;
    int splits;       // >= 1
    int64_t kchunk;   // K elements per split (multiple of BK)
    float* ws;        // split-K slabs [splits][batch][M][N]
    int64_t nbatch;
    int tiles_n;
};

// Stage one operand tile (R rows x BK) from HBM into registers.
// Interior tiles (every row and the whole K step in bounds, 16-B aligned): branch-free 16-B loads that the compiler
// can issue back to back.
template <bool KMAJOR, int R>
__device__ __forceinline__ void load_tile_full(const float* __restrict__ base, int64_t ld, int64_t row0, int64_t k0,
                                               float4 (&reg)[R / 32]) {
    const int t = threadIdx.x;
    if (KMAJOR) {
        const float* p = base + (row0 + (t >> 3)) * ld + k0 + 4 * (t & 7);
#pragma unroll
        for (int j = 0; j < R / 32; ++j) reg[j] = *reinterpret_cast<const float4*>(p + (int64_t)(32 * j) * ld);
    } else {
        constexpr int RV = R / 4;
        constexpr int KSTEP = NTHREADS / RV;
        const float* p = base + (k0 + t / RV) * ld + row0 + 4 * (t % RV);
#pragma unroll
        for (int j = 0; j < R / 32; ++j) reg[j] = *reinterpret_cast<const float4*>(p + (int64_t)(KSTEP * j) * ld);
    }
}

template <bool KMAJOR, int R, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, int64_t ld, int64_t row0, int64_t rows,
                                          int64_t k0, int64_t kend, float4 (&reg)[R / 32]) {
    const int t = threadIdx.x;
    if (KMAJOR) {
        const int c4 = t & 7;
#pragma unroll
        for (int j = 0; j < R / 32; ++j) {
            const int r = (t >> 3) + 32 * j;
            const int64_t gr = row0 + r, gk = k0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < rows) {
                const float* p = base + gr * ld + gk;
                if (VEC && gk + 3 < kend) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (gk + 0 < kend) v.x = p[0];
                    if (gk + 1 < kend) v.y = p[1];
                    if (gk + 2 < kend) v.z = p[2];
                    if (gk + 3 < kend) v.w = p[3];
                }
            }
            reg[j] = v;
        }
    } else {
        constexpr int RV = R / 4;          // float4 per k-row
        constexpr int KSTEP = NTHREADS / RV;
        const int r4 = t % RV;
#pragma unroll
        for (int j = 0; j < R / 32; ++j) {
            const int k = t / RV + KSTEP * j;
            const int64_t gk = k0 + k, gr = row0 + 4 * r4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < kend) {
                const float* p = base + gk * ld + gr;
                if (VEC && gr + 3 < rows) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (gr + 0 < rows) v.x = p[0];
                    if (gr + 1 < rows) v.y = p[1];
                    if (gr + 2 < rows) v.z = p[2];
                    if (gr + 3 < rows) v.w = p[3];
                }
            }
            reg[j] = v;
        }
    }
}

template <bool KMAJOR, int R>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const float4 (&reg)[R / 32]) {
    const int t = threadIdx.x;
    if (KMAJOR) {
        const int c4 = t & 7;
#pragma unroll
        for (int j = 0; j < R / 32; ++j) {
            const int r = (t >> 3) + 32 * j;
            *reinterpret_cast<float4*>(&s[r * LDK + 4 * c4]) = reg[j];
        }
    } else {
        constexpr int RV = R / 4;
        constexpr int KSTEP = NTHREADS / RV;
        const int r4 = t % RV;
#pragma unroll
        for (int j = 0; j < R / 32; ++j) {
            const int k = t / RV + KSTEP * j;
            *reinterpret_cast<float4*>(&s[k * (R + 4) + 4 * r4]) = reg[j];
        }
    }
}

// Fragment for k-chunk c (8 k values): element s of lane half h is k = 8c + 4h + s.
template <bool KMAJOR, int R>
__device__ __forceinline__ void read_frag(const float* __restrict__ s, int roff, int i, int h, int c, float (&f)[4]) {
    if (KMAJOR) {
        const float4 v = *reinterpret_cast<const float4*>(&s[(roff + i) * LDK + 8 * c + 4 * h]);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = s[(8 * c + 4 * h + q) * (R + 4) + roff + i];
    }
}

template <bool TA, bool TB, int BM, int BN, bool VEC>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(const KParams p) {
    constexpr bool AK = !TA;  // A has K contiguous in HBM
    constexpr bool BKM = TB;  // B has K contiguous in HBM
    constexpr int WTM = BM / 64, WTN = BN / 64;  // 32x32 tiles per wave along M / N
    constexpr int SA = BM * LDK, SB = BN * LDK;  // floats per stage (upper bound for both layouts)
    __shared__ __attribute__((aligned(16))) float smem[2 * (SA + SB)];

    // XCD-aware bijective remap of the tile index (blocks with equal blockIdx.x % 8 share an XCD/L2).
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;

    const int z = blockIdx.y;
    const int zb = z / p.splits, ks = z % p.splits;
    const int64_t z1 = zb / p.nb2, z2 = zb % p.nb2;
    const float* A = p.A + z1 * p.sa1 + z2 * p.sa2;
    const float* B = p.B + z1 * p.sb1 + z2 * p.sb2;

    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t kbeg = (int64_t)ks * p.kchunk;
    const int64_t kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
    const int nk = (int)((kend - kbeg + BK - 1) / BK);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int a = 0; a < WTM; ++a)
#pragma unroll
        for (int b = 0; b < WTN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    float4 ra[BM / 32], rb[BN / 32];
    const bool interior = VEC && (m0 + BM <= p.M) && (n0 + BN <= p.N);  // workgroup-uniform
    auto fetch = [&](int64_t k0) {
        if (interior && k0 + BK <= kend) {
            load_tile_full<AK, BM>(A, p.lda, m0, k0, ra);
            load_tile_full<BKM, BN>(B, p.ldb, n0, k0, rb);
        } else {
            load_tile<AK, BM, VEC>(A, p.lda, m0, p.M, k0, kend, ra);
            load_tile<BKM, BN, VEC>(B, p.ldb, n0, p.N, k0, kend, rb);
        }
    };
    if (nk > 0) {
        fetch(kbeg);
        store_tile<AK, BM>(smem, ra);
        store_tile<BKM, BN>(smem + SA, rb);
    }
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const float* sa = smem + cur * (SA + SB);
        const float* sb = sa + SA;
        const bool more = kt + 1 < nk;
        if (more) fetch(kbeg + (int64_t)(kt + 1) * BK);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            float fa[WTM][4], fb[WTN][4];
#pragma unroll
            for (int a = 0; a < WTM; ++a) read_frag<AK, BM>(sa, wm * (BM / 2) + a * 32, i, h, c, fa[a]);
#pragma unroll
            for (int b = 0; b < WTN; ++b) read_frag<BKM, BN>(sb, wn * (BN / 2) + b * 32, i, h, c, fb[b]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < WTM; ++a)
#pragma unroll
                    for (int b = 0; b < WTN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][s], fb[b][s], acc[a][b], 0, 0, 0);
        }
        if (more) {
            float* da = smem + (cur ^ 1) * (SA + SB);
            store_tile<AK, BM>(da, ra);
            store_tile<BKM, BN>(da + SA, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    // Epilogue. C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    if (p.splits > 1) {
        float* W = p.ws + ((int64_t)ks * p.nbatch + zb) * p.M * p.N;
#pragma unroll
        for (int a = 0; a < WTM; ++a)
#pragma unroll
            for (int b = 0; b < WTN; ++b) {
                const int64_t col = n0 + wn * (BN / 2) + b * 32 + i;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = m0 + wm * (BM / 2) + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (row < p.M && col < p.N) W[row * p.N + col] = acc[a][b][e];
                }
            }
    } else if (m0 + BM <= p.M && n0 + BN <= p.N) {  // interior tile: unguarded stores
        float* C = p.C + z1 * p.sc1 + z2 * p.sc2;
#pragma unroll
        for (int a = 0; a < WTM; ++a)
#pragma unroll
            for (int b = 0; b < WTN; ++b) {
                const int64_t col = n0 + wn * (BN / 2) + b * 32 + i;
                const float bv = p.bias ? p.bias[col] : 0.f;
                float* cp = C + (m0 + wm * (BM / 2) + a * 32 + 4 * h) * p.ldc + col;
                if (p.beta != 0.f) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float* q2 = cp + (int64_t)((e & 3) + 8 * (e >> 2)) * p.ldc;
                        *q2 = p.alpha * acc[a][b][e] + p.beta * *q2 + bv;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) cp[(int64_t)((e & 3) + 8 * (e >> 2)) * p.ldc] = p.alpha * acc[a][b][e] + bv;
                }
            }
    } else {
        float* C = p.C + z1 * p.sc1 + z2 * p.sc2;
#pragma unroll
        for (int a = 0; a < WTM; ++a)
#pragma unroll
            for (int b = 0; b < WTN; ++b) {
                const int64_t col = n0 + wn * (BN / 2) + b * 32 + i;
                const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = m0 + wm * (BM / 2) + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (row < p.M && col < p.N) {
                        float v = p.alpha * acc[a][b][e];
                        if (p.beta != 0.f) v += p.beta * C[row * p.ldc + col];
                        C[row * p.ldc + col] = v + bv;
                    }
                }
            }
    }
}

__global__ void splitk_reduce_kernel(const KParams p) {
    const int64_t mn = p.M * p.N;
    const int64_t total = mn * p.nbatch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t zb = idx / mn, rem = idx % mn;
        const int64_t row = rem / p.N, col = rem % p.N;
        float s = 0.f;
        for (int k = 0; k < p.splits; ++k) s += p.ws[((int64_t)k * p.nbatch + zb) * mn + rem];
        const int64_t z1 = zb / p.nb2, z2 = zb % p.nb2;
        float* C = p.C + z1 * p.sc1 + z2 * p.sc2 + row * p.ldc + col;
        float v = p.alpha * s;
        if (p.beta != 0.f) v += p.beta * *C;
        if (p.bias) v += p.bias[col];
        *C = v;
    }
}

struct Plan {
    int bm, bn, splits;
    int64_t kchunk;
    int64_t ws_bytes;
};

Plan make_plan(const dyn_gemm_desc* d) {
    Plan pl;
    const int64_t batch = d->nb1 * d->nb2;
    auto blocks = [&](int bm, int bn) { return dyn::cdiv(d->M, bm) * dyn::cdiv(d->N, bn) * batch; };
    // Largest tile that still gives every CU about two workgroups; otherwise the smallest tile.
    const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    pl.bm = 64; pl.bn = 64;
    for (int c = 0; c < 4; ++c) {
        if (blocks(cand[c][0], cand[c][1]) >= 448) { pl.bm = cand[c][0]; pl.bn = cand[c][1]; break; }
    }
    if (d->M <= 64 && pl.bm == 128) pl.bm = 64;
    if (d->N <= 64 && pl.bn == 128) pl.bn = 64;
    int splits = d->split_k;
    const int64_t nb = blocks(pl.bm, pl.bn);
    if (splits == 0) {  // auto: only when the grid cannot fill the chip and K is deep
        splits = 1;
        if (nb < 256 && d->K >= 1024) {
            int64_t want = dyn::cdiv(512, nb), maxs = d->K / 256;
            splits = (int)(want < maxs ? want : maxs);
            if (splits < 1) splits = 1;
            if (splits > 32) splits = 32;
        }
    }
    if (splits < 1) splits = 1;
    int64_t ktiles = dyn::cdiv(d->K, BK);
    if (splits > ktiles) splits = (int)(ktiles > 0 ? ktiles : 1);
    pl.kchunk = dyn::cdiv(ktiles, splits) * BK;
    splits = (int)dyn::cdiv(d->K > 0 ? d->K : 1, pl.kchunk);
    pl.splits = splits;
    pl.ws_bytes = splits > 1 ? (int64_t)splits * batch * d->M * d->N * (int64_t)sizeof(float) : 0;
    return pl;
}

template <bool TA, bool TB, int BM, int BN>
void launch_vec(const KParams& kp, bool vec, dim3 grid, hipStream_t st) {
    if (vec) hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, BM, BN, true>), grid, dim3(NTHREADS), 0, st, kp);
    else hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, BM, BN, false>), grid, dim3(NTHREADS), 0, st, kp);
}

template <bool TA, bool TB>
void launch_tile(const KParams& kp, const Plan& pl, bool vec, dim3 grid, hipStream_t st) {
    if (pl.bm == 128 && pl.bn == 128) launch_vec<TA, TB, 128, 128>(kp, vec, grid, st);
    else if (pl.bm == 128 && pl.bn == 64) launch_vec<TA, TB, 128, 64>(kp, vec, grid, st);
    else if (pl.bm == 64 && pl.bn == 128) launch_vec<TA, TB, 64, 128>(kp, vec, grid, st);
    else launch_vec<TA, TB, 64, 64>(kp, vec, grid, st);
}

}  // namespace

extern "C" int64_t dyn_gemm_f32_workspace_bytes(const dyn_gemm_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->nb1 <= 0 || d->nb2 <= 0) return 0;
    return make_plan(d).ws_bytes;
}

extern "C" int dyn_gemm_f32(const dyn_gemm_desc* d, void* stream) {
    DYN_REQUIRE(d != nullptr, DYN_E_ARG, "dyn_gemm_f32: null descriptor");
    DYN_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0 && d->nb1 >= 1 && d->nb2 >= 1, DYN_E_ARG,
                "dyn_gemm_f32: bad sizes M=%lld N=%lld K=%lld nb=%lldx%lld", (long long)d->M, (long long)d->N,
                (long long)d->K, (long long)d->nb1, (long long)d->nb2);
    if (d->M == 0 || d->N == 0) return DYN_OK;
    DYN_REQUIRE(d->A && d->B && d->C, DYN_E_ARG, "dyn_gemm_f32: null operand");
    DYN_REQUIRE(d->lda >= (d->trans_a ? d->M : d->K) && d->ldb >= (d->trans_b ? d->K : d->N) && d->ldc >= d->N,
                DYN_E_ARG, "dyn_gemm_f32: leading dimension smaller than the row length");
    Plan pl = make_plan(d);
    if (pl.splits > 1 && (d->workspace == nullptr || d->workspace_bytes < pl.ws_bytes)) {
        DYN_REQUIRE(d->split_k == 0, DYN_E_WORKSPACE, "dyn_gemm_f32: split_k=%d needs %lld workspace bytes, got %lld",
                    d->split_k, (long long)pl.ws_bytes, (long long)d->workspace_bytes);
        pl.splits = 1;  // auto split silently degrades to a single pass
        pl.kchunk = dyn::cdiv(d->K > 0 ? d->K : 1, BK) * BK;
        pl.ws_bytes = 0;
    }
    const int64_t batch = d->nb1 * d->nb2;
    KParams kp;
    kp.M = d->M; kp.N = d->N; kp.K = d->K;
    kp.A = d->A; kp.lda = d->lda; kp.sa1 = d->sa1; kp.sa2 = d->sa2;
    kp.B = d->B; kp.ldb = d->ldb; kp.sb1 = d->sb1; kp.sb2 = d->sb2;
    kp.C = d->C; kp.ldc = d->ldc; kp.sc1 = d->sc1; kp.sc2 = d->sc2;
    kp.bias = d->bias; kp.alpha = d->alpha; kp.beta = d->beta;
    kp.nb2 = d->nb2; kp.splits = pl.splits; kp.kchunk = pl.kchunk;
    kp.ws = (float*)d->workspace; kp.nbatch = batch;
    const int64_t tiles_m = dyn::cdiv(d->M, pl.bm), tiles_n = dyn::cdiv(d->N, pl.bn);
    kp.tiles_n = (int)tiles_n;
    DYN_REQUIRE(tiles_m * tiles_n < (1ll << 31) && batch * pl.splits < 65536, DYN_E_ARG,
                "dyn_gemm_f32: grid too large (tiles=%lld, z=%lld)", (long long)(tiles_m * tiles_n),
                (long long)(batch * pl.splits));
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    const bool vec = al16(d->A) && al16(d->B) && (d->lda % 4 == 0) && (d->ldb % 4 == 0) && (d->sa1 % 4 == 0) &&
                     (d->sa2 % 4 == 0) && (d->sb1 % 4 == 0) && (d->sb2 % 4 == 0);
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)(batch * pl.splits));
    hipStream_t st = (hipStream_t)stream;
    if (!d->trans_a && !d->trans_b) launch_tile<false, false>(kp, pl, vec, grid, st);
    else if (!d->trans_a && d->trans_b) launch_tile<false, true>(kp, pl, vec, grid, st);
    else if (d->trans_a && !d->trans_b) launch_tile<true, false>(kp, pl, vec, grid, st);
    else launch_tile<true, true>(kp, pl, vec, grid, st);
    int rc = dyn::check_launch("dyn_gemm_f32");
    if (rc != DYN_OK) return rc;
    if (pl.splits > 1) {
        const int64_t total = d->M * d->N * batch;
        int64_t nblk = dyn::cdiv(total, 256);
        if (nblk > 2048) nblk = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, st, kp);
        rc = dyn::check_launch("dyn_gemm_f32(splitk_reduce)");
    }
    return rc;
}
