// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-exact fmaf chain).
//
// Replaces the dense products inside model(audio_signal=...) and loss.backward() of the reference's
// dynamic-eval loop (reference lcasr/lib.py:550,579,603): QKV/out projections, FFN, pointwise convs,
// subsampling projections, CTC head, self-conditioning re-projection, and every dgrad/wgrad of those.
//
// Design (MI355X-first, not a warp-tiled port):
//   * 256-thread workgroup = 4 wave64 as 2(M) x 2(N); each wave owns (BM/2)x(BN/2) as 32x32 MFMA tiles,
//     accumulators stay in registers for the whole K loop (64 AGPRs at 128x128).  BK = 32.
//   * Staging, main path: direct-to-LDS loads (global_load_lds_dwordx4, GldsStager) — no staging VGPRs, no ds_write pass.
//     One wave-instruction writes 1 KiB of LDS linearly, so the image is unpadded and the conflict-free layout comes from
//     the SOURCE address each lane fetches (K-contiguous operand: 16-B slot q of row r stored at q ^ ((r >> 1) & 7);
//     row-contiguous operand: [k][rows], read with ds_read_b32).  Whole K tiles and 16-B aligned operands only; matrix
//     edges are clamped loads.
//   * Staging, fallback (partial K tiles, unaligned operands, lda < K implicit GEMMs): global -> registers -> LDS with
//     the padded [row][BK+4] image (one ds_read_b128 per lane feeds FOUR k-steps: lane half h covers k = 8c+4h+s).
//     Both paths feed the MFMAs the same operands in the same order: bit-identical results.
//     DYN_GEMM_GLDS=0 in the environment forces the fallback everywhere (A/B measurements).
//   * Two-level software pipeline: fragments double-buffered in registers, a tile's last chunk issued after the barrier
//     and after the next tile's first fragment reads; one barrier per K tile.
//   * XCD-aware bijective block remap: the 8 XCDs (private L2s) each walk a contiguous run of output tiles.
//   * split-K writes fp32 slabs to a caller workspace and a second kernel sums them in slice order; the last partial
//     round of workgroups is K-sliced the same way (tail slicing): deterministic, no float atomics.
#include <algorithm>
#include "common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

#ifndef DYN_GEMM_BK
#define DYN_GEMM_BK 32
#endif
constexpr int BK = DYN_GEMM_BK;
constexpr int LDK = BK + 4;
constexpr int KV = BK / 4;            // float4 per staged row when K is contiguous
constexpr int RP = 256 / KV;          // rows staged per pass of the 256 threads
constexpr int NTHREADS = 256;

struct KParams {
    int64_t M, N, K;
    const float* A; int64_t lda, sa1, sa2;
    const float* B; int64_t ldb, sb1, sb2;
    float* C; int64_t ldc, sc1, sc2;
    int64_t cin_delta;  // C_in - C in elements (0 = accumulate in place): beta reads the residual from another buffer
    const float* bias;
    int64_t bs1, bs2;    // batch strides of the bias (0, 0: one bias for every batch; a weight-batched product has one bias row per weight)
    float alpha, beta;
    int64_t nb2;
    int splits;       // >= 1
    int64_t kchunk;   // K elements per split (multiple of BK)
    float* ws;        // split-K slabs [splits][batch][M][N]
    int64_t nbatch;
    int tiles_n;
    int tiles_m;
    int n_major;      // 1: consecutive items walk down a COLUMN of tiles (each XCD's contiguous run then shares B panels and
                      // partitions A instead of the other way round): chosen when B is the larger operand
    int64_t tiles_per_batch;
    int64_t full_items;  // work items (batch, tile, k-split) handled by one workgroup each; the rest is the TAIL
    int tail_f;          // every tail item is cut into tail_f K-slices so the last partial round of workgroups still
    float* tail_ws;      // fills the chip; slices go to tile-local slabs [tail item][slice][BM][BN] and are summed in order
    // epilogue extras (interior, non-split path only)
    int epi;             // DYN_GEMM_EPI_*: 0 none; 1 C = silu(v), aux = v (pre-activation kept for the backward); 2 C = v * silu'(aux)
    float* aux;          // addressed like C (same ldc / batch strides)
    // grouped launches only
    int* counters;       // arrival counters of the partial tiles (zero before and after every launch); NULL = separate reduce kernels
    float* colsum;       // [M]: colsum[m] = colsum_beta * colsum[m] + sum_k A(m, k)   (bias gradient of a weight-gradient GEMM)
    float colsum_beta;
    int64_t first_item;  // first work item of this group in the grouped launch
};

// alpha * acc + beta * c_in + bias in ONE pinned operation order (explicit mul / fma / add: no contraction differences between the
// tile epilogue and the reduce kernels, so every plan of a product rounds its outputs the same way)
__device__ __forceinline__ float epi_value(float alpha, float acc, float beta, const float* cin, float bias) {
    float v = __fmul_rn(alpha, acc);
    if (beta != 0.f) v = __fmaf_rn(beta, *cin, v);
    return __fadd_rn(v, bias);
}

// same formulas as silu_fwd_kernel / silu_bwd_kernel (elementwise.hip): fused and unfused paths agree bit for bit
__device__ __forceinline__ float silu_f(float v) { return v * dyn::sigmoidf_(v); }
__device__ __forceinline__ float silu_grad_f(float u) {
    const float sg = dyn::sigmoidf_(u);
    return sg * (1.f + u * (1.f - sg));
}

// Stage one operand tile (R rows x BK) from HBM into registers.
// Interior tiles (every row and the whole K step in bounds, 16-B aligned): branch-free 16-B loads that the compiler
// can issue back to back.
template <bool KMAJOR, int R>
__device__ __forceinline__ void load_tile_full(const float* __restrict__ base, int64_t ld, int64_t row0, int64_t k0,
                                               float4 (&reg)[(R * BK / 1024)]) {
    const int t = threadIdx.x;
    if (KMAJOR) {
        const float* p = base + (row0 + (t / KV)) * ld + k0 + 4 * (t % KV);
#pragma unroll
        for (int j = 0; j < (R * BK / 1024); ++j) reg[j] = *reinterpret_cast<const float4*>(p + (int64_t)(RP * j) * ld);
    } else {
        constexpr int RV = R / 4;
        constexpr int KSTEP = NTHREADS / RV;
        const float* p = base + (k0 + t / RV) * ld + row0 + 4 * (t % RV);
#pragma unroll
        for (int j = 0; j < (R * BK / 1024); ++j) reg[j] = *reinterpret_cast<const float4*>(p + (int64_t)(KSTEP * j) * ld);
    }
}

template <bool KMAJOR, int R, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, int64_t ld, int64_t row0, int64_t rows,
                                          int64_t k0, int64_t kend, float4 (&reg)[(R * BK / 1024)]) {
    const int t = threadIdx.x;
    if (KMAJOR) {
        const int c4 = t % KV;
#pragma unroll
        for (int j = 0; j < (R * BK / 1024); ++j) {
            const int r = (t / KV) + RP * j;
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
        for (int j = 0; j < (R * BK / 1024); ++j) {
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
__device__ __forceinline__ void store_tile(float* __restrict__ s, const float4 (&reg)[(R * BK / 1024)]) {
    const int t = threadIdx.x;
    if (KMAJOR) {
        const int c4 = t % KV;
#pragma unroll
        for (int j = 0; j < (R * BK / 1024); ++j) {
            const int r = (t / KV) + RP * j;
            *reinterpret_cast<float4*>(&s[r * LDK + 4 * c4]) = reg[j];
        }
    } else {
        constexpr int RV = R / 4;
        constexpr int KSTEP = NTHREADS / RV;
        const int r4 = t % RV;
#pragma unroll
        for (int j = 0; j < (R * BK / 1024); ++j) {
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

// ---- direct-to-LDS staging (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass --------------------------------
// One wave-instruction writes 1 KiB of LDS linearly (wave-uniform base + lane * 16 B), so the LDS image is unpadded and the
// bank-conflict-free layout is obtained by choosing which 16 B each lane FETCHES:
//   K-contiguous operand: [row][32 floats]; the 16-B slot q (= k / 4) of row r lives at slot q ^ ((r >> 1) & 7), which puts
//     every 16-lane ds_read_b128 group on 16 distinct slots of the 256-B bank row;
//   row-contiguous operand: [k][R floats], read with ds_read_b32 (32 consecutive floats per lane half: conflict-free as is).
// Rows past the matrix edge are clamped to the last valid row (group of 4): they only feed outputs that are never stored.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <bool KMAJOR, int R, int NW = 4>
struct GldsStager {
    static constexpr int NI = R * BK / (256 * NW);   // 1-KiB pieces per wave per tile: 4 waves: R=128 -> 4, R=64 -> 2; 8 waves: R=256 -> 4, R=128 -> 2
    const float* ptr[NI];
    int64_t step;
    __device__ __forceinline__ void init(const float* base, int64_t ld, int64_t row0, int64_t rows, int64_t k0, int wave, int lane) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = wave * NI + j;
            if (KMAJOR) {
                const int row = n * 8 + (lane >> 3);
                const int q = (lane & 7) ^ ((row >> 1) & 7);
                int64_t gr = row0 + row;
                gr = gr < rows ? gr : rows - 1;
                ptr[j] = base + gr * ld + k0 + 4 * q;
            } else {
                const int o = n * 256 + lane * 4;
                const int k = o / R, col = o % R;
                int64_t gr = row0 + col;
                gr = gr < rows ? gr : rows - 4;
                ptr[j] = base + (k0 + k) * ld + gr;
            }
        }
        step = KMAJOR ? (int64_t)BK : (int64_t)BK * ld;
    }
    __device__ __forceinline__ void issue(float* tile, int wave) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            __builtin_amdgcn_global_load_lds((gbl_void_t*)ptr[j], (lds_void_t*)(tile + (wave * NI + j) * 256), 16, 0, 0);
            ptr[j] += step;
        }
    }
};

template <bool KMAJOR, int R>
__device__ __forceinline__ void read_frag_glds(const float* __restrict__ s, int roff, int i, int h, int c, float (&f)[4]) {
    if (KMAJOR) {
        const float4 v = *reinterpret_cast<const float4*>(&s[(roff + i) * BK + 4 * ((2 * c + h) ^ ((i >> 1) & 7))]);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = s[(8 * c + 4 * h + q) * R + roff + i];
    }
}

// One output tile (or one K slice of it) by one 256-thread workgroup.  `bid` = position of this workgroup among the launch's work
// items (the single-GEMM kernel passes blockIdx.x; the grouped kernel passes the index inside the group, already remapped).
// NW = waves per workgroup: 4 (2 x 2 over the tile) or 8 (4 x 2: the 256x128 tile, direct-to-LDS staging only — a wave's share of the tile is
// 64x64 as in the 128x128 tile, but the workgroup moves 25 % fewer operand bytes per MFMA through the L2 -> LDS path).
template <bool TA, bool TB, int BM, int BN, bool VEC, bool GLDS, bool GROUPED, int NW = 4>
__device__ __forceinline__ void gemm_tile(const KParams& p, const int64_t bid, float* smem) {
    constexpr bool AK = !TA;  // A has K contiguous in HBM
    constexpr bool BKM = TB;  // B has K contiguous in HBM
    constexpr int WM = NW / 2;                          // waves along M (2 along N)
    constexpr int WTM = BM / (32 * WM), WTN = BN / 64;  // 32x32 tiles per wave along M / N
    static_assert(NW == 4 || (NW == 8 && GLDS), "the 8-wave tile exists for direct-to-LDS staging only");
    static_assert(!GLDS || (VEC && BK == 32), "direct-to-LDS staging needs 16-B aligned operands and BK = 32");
    constexpr int SA = GLDS ? BM * BK : BM * LDK, SB = GLDS ? BN * BK : BN * LDK;  // floats per stage

    // Work item of this workgroup.  Full items: XCD-aware bijective remap (blocks with equal blockIdx.x % 8 share an
    // XCD/L2, so each XCD walks a contiguous run of tiles).  Tail items come last in dispatch order and are K-sliced.
    const bool is_tail = !GROUPED && bid >= p.full_items;
    int64_t item;
    int sub = 0;
    if (GROUPED) {
        item = bid;          // the grouped kernel remaps over the whole launch before it picks the group
    } else if (!is_tail) {
        const int64_t nwg = p.full_items;
        const int64_t xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    } else {
        const int64_t t = bid - p.full_items;
        item = p.full_items + t / p.tail_f;
        sub = (int)(t % p.tail_f);
    }
    const int ks = (int)(item % p.splits);
    const int64_t t2 = item / p.splits;
    const int tile = (int)(t2 % p.tiles_per_batch);
    const int zb = (int)(t2 / p.tiles_per_batch);
    const int tm = p.n_major ? tile % p.tiles_m : tile / p.tiles_n;
    const int tn = p.n_major ? tile / p.tiles_m : tile % p.tiles_n;
    const int64_t z1 = zb / p.nb2, z2 = zb % p.nb2;
    const float* A = p.A + z1 * p.sa1 + z2 * p.sa2;
    const float* B = p.B + z1 * p.sb1 + z2 * p.sb2;

    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    int64_t kbeg = (int64_t)ks * p.kchunk;
    int64_t kend = (kbeg + p.kchunk < p.K) ? kbeg + p.kchunk : p.K;
    if (is_tail) {
        const int64_t kt_all = (kend - kbeg + BK - 1) / BK;
        const int64_t per = (kt_all + p.tail_f - 1) / p.tail_f;
        const int64_t kb2 = kbeg + (int64_t)sub * per * BK;
        const int64_t ke2 = kb2 + per * BK;
        kend = ke2 < kend ? ke2 : kend;
        kbeg = kb2;
    }
    const int nk = kend > kbeg ? (int)((kend - kbeg + BK - 1) / BK) : 0;

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

    float4 ra[BM * BK / 1024], rb[BN * BK / 1024];
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
    GldsStager<AK, BM, NW> stA;
    GldsStager<BKM, BN, NW> stB;
    if (GLDS) {
        stA.init(A, p.lda, m0, p.M, kbeg, wave, lane);
        stB.init(B, p.ldb, n0, p.N, kbeg, wave, lane);
        if (nk > 0) { stA.issue(smem, wave); stB.issue(smem + SA, wave); }
    } else if (nk > 0) {
        fetch(kbeg);
        store_tile<AK, BM>(smem, ra);
        store_tile<BKM, BN>(smem + SA, rb);
    }
    // An LDS-DMA is ordered for a ds_read only by the issuing wave's vmcnt wait followed by a barrier: state the wait explicitly
    // instead of relying on the fence hipcc attaches to __syncthreads().
    if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Main loop, software-pipelined at two levels so a wave's MFMA stream never waits on LDS latency:
    //   * fragments are double-buffered in registers: chunk c+1 is read from LDS while chunk c's MFMAs issue;
    //   * the tile's LAST chunk is issued AFTER the barrier and after the next tile's first fragment reads, so the
    //     ds_write + barrier + first ds_read latency of tile t+1 hides behind MFMAs of tile t.
    // One barrier per K tile: tile t+1 is written to the other LDS buffer before it, and every read of tile t's buffer
    // (the last chunk's fragments included) has been issued and waited for before it.
    constexpr int NC = BK / 8;
    constexpr int NREADS = (AK ? WTM : 4 * WTM) + (BKM ? WTN : 4 * WTN);   // LDS read instructions per chunk
    static_assert(NC % 2 == 0, "the register ping-pong assumes an even number of k-chunks per tile");
    float fa[2][WTM][4], fb[2][WTN][4];
    auto read_chunk = [&](const float* sa, const float* sb, int c, int slot) {
#pragma unroll
        for (int a = 0; a < WTM; ++a) {
            if (GLDS) read_frag_glds<AK, BM>(sa, wm * (BM / WM) + a * 32, i, h, c, fa[slot][a]);
            else read_frag<AK, BM>(sa, wm * (BM / WM) + a * 32, i, h, c, fa[slot][a]);
        }
#pragma unroll
        for (int b = 0; b < WTN; ++b) {
            if (GLDS) read_frag_glds<BKM, BN>(sb, wn * (BN / 2) + b * 32, i, h, c, fb[slot][b]);
            else read_frag<BKM, BN>(sb, wn * (BN / 2) + b * 32, i, h, c, fb[slot][b]);
        }
    };
    // grouped weight-gradient launches: the first column of tiles also sums its A panel over K (the bias gradient), from the
    // fragments that are in registers anyway; waves wn == 0 only (both wn read the same A fragments)
    float csum[WTM];
#pragma unroll
    for (int a = 0; a < WTM; ++a) csum[a] = 0.f;
    const bool do_colsum = GROUPED && p.colsum != nullptr && tn == 0 && wn == 0;
    auto mfma_chunk = [&](int slot) {
        if (GROUPED && do_colsum) {
#pragma unroll
            for (int a = 0; a < WTM; ++a) csum[a] += (fa[slot][a][0] + fa[slot][a][1]) + (fa[slot][a][2] + fa[slot][a][3]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < WTM; ++a)
#pragma unroll
                for (int b = 0; b < WTN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][a][s], fb[slot][b][s], acc[a][b], 0, 0, 0);
    };
    int cur = 0;
    if (nk > 0) read_chunk(smem, smem + SA, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const float* sa = smem + cur * (SA + SB);
        const float* sb = sa + SA;
        const bool more = kt + 1 < nk;
        float* da = smem + (cur ^ 1) * (SA + SB);
        if (GLDS) {
            // the other buffer is free: its last fragment reads were waited for before the previous barrier
            if (more) { stA.issue(da, wave); stB.issue(da + SA, wave); }
        } else if (more) {
            fetch(kbeg + (int64_t)(kt + 1) * BK);
        }
#pragma unroll
        for (int c = 0; c < NC - 1; ++c) {
            read_chunk(sa, sb, c + 1, (c + 1) & 1);
            mfma_chunk(c & 1);
            // pin the order the source states (LDS reads of the next chunk first, then this chunk's MFMAs): left alone,
            // the scheduler sinks the reads to just before their first use and the MFMA stream stalls on LDS latency
            __builtin_amdgcn_sched_group_barrier(0x100, NREADS, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * WTM * WTN, 0);
        }
        if (!GLDS && more) {
            store_tile<AK, BM>(da, ra);
            store_tile<BKM, BN>(da + SA, rb);
        }
        if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of tile kt+1 has landed (see the prologue)
        __syncthreads();
        if (more) read_chunk(da, da + SA, 0, 0);
        mfma_chunk((NC - 1) & 1);
        __builtin_amdgcn_sched_group_barrier(0x100, NREADS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * WTM * WTN, 0);
        cur ^= 1;
    }

    if (GROUPED && do_colsum) {   // lane halves hold the two k sub-ranges of every chunk: fold them, then lanes h == 0 own row i
#pragma unroll
        for (int a = 0; a < WTM; ++a) {
            const float tot = csum[a] + __shfl_xor(csum[a], 32);
            const int64_t m = m0 + wm * (BM / WM) + a * 32 + i;
            if (h == 0 && m < p.M) p.colsum[m] = (p.colsum_beta != 0.f ? p.colsum_beta * p.colsum[m] : 0.f) + tot;
        }
    }
    // Epilogue. C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    const bool partial = !GROUPED && (is_tail || p.splits > 1);
    if (partial && p.counters == nullptr) {       // legacy: slabs now, a second kernel sums them
        if (is_tail) {
            float* W = p.tail_ws + ((item - p.full_items) * p.tail_f + sub) * (int64_t)(BM * BN);
#pragma unroll
            for (int a = 0; a < WTM; ++a)
#pragma unroll
                for (int b = 0; b < WTN; ++b) {
                    float* wp = W + (wm * (BM / WM) + a * 32 + 4 * h) * BN + wn * (BN / 2) + b * 32 + i;
#pragma unroll
                    for (int e = 0; e < 16; ++e) wp[((e & 3) + 8 * (e >> 2)) * BN] = acc[a][b][e];
                }
        } else {
            float* W = p.ws + ((int64_t)ks * p.nbatch + zb) * p.M * p.N;
#pragma unroll
            for (int a = 0; a < WTM; ++a)
#pragma unroll
                for (int b = 0; b < WTN; ++b) {
                    const int64_t col = n0 + wn * (BN / 2) + b * 32 + i;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int64_t row = m0 + wm * (BM / WM) + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (row < p.M && col < p.N) W[row * p.N + col] = acc[a][b][e];
                    }
                }
        }
        return;
    }
    if (partial) {
        // K slice `sl` of `nsl` of partial tile `pt`: store the slab, take a ticket; the workgroup that arrives LAST sums all
        // slabs of the tile in slice order (fixed order: the result does not depend on who is last) and runs the epilogue.
        const int64_t pt = is_tail ? (item - p.full_items) : t2;
        const int sl = is_tail ? sub : ks;
        const int nsl = is_tail ? p.tail_f : p.splits;
        float* W0 = p.tail_ws + pt * nsl * (int64_t)(BM * BN);
        float* W = W0 + (int64_t)sl * (BM * BN);
#pragma unroll
        for (int a = 0; a < WTM; ++a)
#pragma unroll
            for (int b = 0; b < WTN; ++b) {
                float* wp = W + (wm * (BM / WM) + a * 32 + 4 * h) * BN + wn * (BN / 2) + b * 32 + i;
#pragma unroll
                for (int e = 0; e < 16; ++e) wp[((e & 3) + 8 * (e >> 2)) * BN] = acc[a][b][e];
            }
        __threadfence();                       // release the slab at device scope (other XCDs have other L2s)
        __syncthreads();                       // every wave's stores are fenced; all LDS fragment reads are long done
        int* flag = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) flag[0] = (atomicAdd(&p.counters[pt], 1) == nsl - 1) ? 1 : 0;
        __syncthreads();
        if (flag[0] == 0) return;
        __threadfence();                       // acquire the other slices' slabs
#pragma unroll
        for (int a = 0; a < WTM; ++a)
#pragma unroll
            for (int b = 0; b < WTN; ++b) {
                const float* rp = W0 + (wm * (BM / WM) + a * 32 + 4 * h) * BN + wn * (BN / 2) + b * 32 + i;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float* q = rp + ((e & 3) + 8 * (e >> 2)) * BN;
                    float t = q[0];
                    for (int k = 1; k < nsl; ++k) t += q[(int64_t)k * (BM * BN)];
                    acc[a][b][e] = t;
                }
            }
        if (threadIdx.x == 0) p.counters[pt] = 0;     // ready for the next launch on this stream
    }
    if (m0 + BM <= p.M && n0 + BN <= p.N) {  // interior tile: unguarded stores
        float* C = p.C + z1 * p.sc1 + z2 * p.sc2;
#pragma unroll
        for (int a = 0; a < WTM; ++a)
#pragma unroll
            for (int b = 0; b < WTN; ++b) {
                const int64_t col = n0 + wn * (BN / 2) + b * 32 + i;
                const float bv = p.bias ? p.bias[z1 * p.bs1 + z2 * p.bs2 + col] : 0.f;
                float* cp = C + (m0 + wm * (BM / WM) + a * 32 + 4 * h) * p.ldc + col;
                if (p.epi != 0) {             // activation fused into the store (host guarantees: no split / tail, aux set)
                    const int64_t ad = p.aux - p.C;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float* q2 = cp + (int64_t)((e & 3) + 8 * (e >> 2)) * p.ldc;
                        const float v = epi_value(p.alpha, acc[a][b][e], p.beta, q2 + p.cin_delta, bv);
                        if (p.epi == 1) { if (p.aux) q2[ad] = v; *q2 = silu_f(v); }
                        else *q2 = v * silu_grad_f(q2[ad]);
                    }
                } else if (p.beta != 0.f) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float* q2 = cp + (int64_t)((e & 3) + 8 * (e >> 2)) * p.ldc;
                        *q2 = epi_value(p.alpha, acc[a][b][e], p.beta, q2 + p.cin_delta, bv);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) cp[(int64_t)((e & 3) + 8 * (e >> 2)) * p.ldc] = epi_value(p.alpha, acc[a][b][e], 0.f, nullptr, bv);
                }
            }
    } else {
        float* C = p.C + z1 * p.sc1 + z2 * p.sc2;
#pragma unroll
        for (int a = 0; a < WTM; ++a)
#pragma unroll
            for (int b = 0; b < WTN; ++b) {
                const int64_t col = n0 + wn * (BN / 2) + b * 32 + i;
                const float bv = (p.bias && col < p.N) ? p.bias[z1 * p.bs1 + z2 * p.bs2 + col] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = m0 + wm * (BM / WM) + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (row < p.M && col < p.N) {
                        float v = epi_value(p.alpha, acc[a][b][e], p.beta, C + row * p.ldc + col + p.cin_delta, bv);
                        if (p.epi == 1) { if (p.aux) C[row * p.ldc + col + (p.aux - p.C)] = v; v = silu_f(v); }
                        else if (p.epi == 2) v *= silu_grad_f(C[row * p.ldc + col + (p.aux - p.C)]);
                        C[row * p.ldc + col] = v;
                    }
                }
            }
    }
}

// amdgpu_waves_per_eu(2, 8): without it hipcc spends the whole 512-register budget of a 4-wave workgroup on the 128x128 tile (252
// per wave: two such waves fill a SIMD's register file and NOTHING else can be resident beside them); asked for two waves per SIMD it
// needs 185, which leaves room for a third, lighter wave — the HBM-bound kernels of the other recording chains.
#ifndef DYN_GEMM_WAVES_PER_EU
#define DYN_GEMM_WAVES_PER_EU 2
#endif
template <bool TA, bool TB, int BM, int BN, bool VEC, bool GLDS, int NW = 4>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(DYN_GEMM_WAVES_PER_EU, 8))) void gemm_f32_kernel(const KParams p) {
    constexpr int SA = GLDS ? BM * BK : BM * LDK, SB = GLDS ? BN * BK : BN * LDK;
    __shared__ __attribute__((aligned(16))) float smem[2 * (SA + SB)];       // 64 KB at 128x128, 96 KB at 256x128 (160 KB of LDS per CU on gfx950)
    gemm_tile<TA, TB, BM, BN, VEC, GLDS, false, NW>(p, (int64_t)blockIdx.x, smem);
}

// Grouped launch: ONE grid over the tiles of up to kMaxGroups independent GEMMs that share (TA, TB) and the tile shape (the
// deferred weight-gradient products of a whole backward pass: each alone has 36-144 tiles of 128x128, far too few for 256 CUs).
// The descriptor table lives in device memory (written by gemm_table_kernel from kernel arguments, so the pair is capturable in
// a hipGraph); a workgroup remaps its index XCD-wise over the whole launch, finds its group by binary search over first_item
// and runs the ordinary tile body.  No split-K and no tail slicing: the launch has thousands of tiles.
constexpr int kMaxGroups = 96;
constexpr int kTableChunk = 12;
struct KTableChunk { KParams g[kTableChunk]; };

__global__ void gemm_table_kernel(const KTableChunk chunk, KParams* __restrict__ table, int base, int count) {
    const int t = threadIdx.x;
    if (t < count) table[base + t] = chunk.g[t];
}

template <bool TA, bool TB, int BM, int BN, bool GLDS>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_grouped_kernel(const KParams* __restrict__ table, int n_groups, int64_t total_items) {
    constexpr int SA = GLDS ? BM * BK : BM * LDK, SB = GLDS ? BN * BK : BN * LDK;
    __shared__ __attribute__((aligned(16))) float smem[2 * (SA + SB)];
    const int64_t bid = blockIdx.x;
    const int64_t xcd = bid & 7, q = total_items >> 3, r = total_items & 7;
    const int64_t item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    int lo = 0, hi = n_groups - 1;            // last group whose first_item <= item
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_item <= item) lo = mid; else hi = mid - 1;
    }
    const int g = __builtin_amdgcn_readfirstlane(lo);
    const KParams p = table[g];
    gemm_tile<TA, TB, BM, BN, true, GLDS, true>(p, item - p.first_item, smem);
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
        float v = epi_value(p.alpha, s, p.beta, C + p.cin_delta, p.bias ? p.bias[z1 * p.bs1 + z2 * p.bs2 + col] : 0.f);
        if (p.epi == 1) { if (p.aux) C[p.aux - p.C] = v; v = silu_f(v); }
        else if (p.epi == 2) v *= silu_grad_f(C[p.aux - p.C]);
        *C = v;
    }
}

// C tile = alpha * sum_slices slab + beta * C + bias for every tail item (slices added in order: deterministic).
__global__ __launch_bounds__(256) void tail_reduce_kernel(const KParams p, int BM, int BN, int64_t tail_items) {
    const int per_tile = BM * BN / 4;  // float4 per tile
    const int64_t total = tail_items * per_tile;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ti = idx / per_tile;
        const int e4 = (int)(idx % per_tile);
        const int r = (e4 * 4) / BN, c = (e4 * 4) % BN;
        const float4* W = reinterpret_cast<const float4*>(p.tail_ws + (ti * p.tail_f) * (int64_t)(BM * BN)) + e4;
        float4 s4 = W[0];
        for (int k = 1; k < p.tail_f; ++k) {
            const float4 v = W[(int64_t)k * per_tile];
            s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
        }
        const int64_t item = p.full_items + ti;  // splits == 1 in tail mode
        const int tile = (int)(item % p.tiles_per_batch);
        const int zb = (int)(item / p.tiles_per_batch);
        const int64_t row = (int64_t)(p.n_major ? tile % p.tiles_m : tile / p.tiles_n) * BM + r;
        const int64_t col0 = (int64_t)(p.n_major ? tile / p.tiles_m : tile % p.tiles_n) * BN + c;
        if (row >= p.M) continue;
        const int64_t z1 = zb / p.nb2, z2 = zb % p.nb2;
        float* C = p.C + z1 * p.sc1 + z2 * p.sc2 + row * p.ldc;
        const float v[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t col = col0 + j;
            if (col < p.N) {
                float o = epi_value(p.alpha, v[j], p.beta, C + col + p.cin_delta, p.bias ? p.bias[z1 * p.bs1 + z2 * p.bs2 + col] : 0.f);
                if (p.epi == 1) { if (p.aux) C[col + (p.aux - p.C)] = o; o = silu_f(o); }
                else if (p.epi == 2) o *= silu_grad_f(C[col + (p.aux - p.C)]);
                C[col] = o;
            }
        }
    }
}

struct Plan {
    int bm, bn, splits;
    int64_t kchunk;
    int64_t ws_bytes;       // global split-K slabs
    int64_t full_items;
    int64_t tail_items;
    int tail_f;
    int64_t tail_ws_bytes;
};

// Tile / split selection by a small cost model (all shapes of this path are known up front: M in {2048, 4096, 8192,...},
// N, K in {128 ... 4096}).  A "round" is one workgroup per slot (slots = resident workgroups per CU x 256 CUs); a partial
// last round is the classic wave-quantisation loss, so when no global split-K is used its items are K-sliced tail_f ways
// (tail_f = slots / remainder) and run as one short, full round.
double tile_eff(int bm, int bn) { return bm == 256 ? 0.84 : (bm == 128 && bn == 128) ? 0.80 : (bm == 64 && bn == 64) ? 0.62 : 0.72; }
int tile_occ(int bm, int bn) { return bm == 256 ? 1 : (bm == 64 && bn == 64) ? 4 : (bm == 128 && bn == 128) ? 2 : 3; }  // resident workgroups per CU (LDS-bound)

// direct-to-LDS staging: whole K tiles only (no zero fill), 16-B aligned operands, and an operand whose rows are the contiguous axis
// must have a row count that is a multiple of 4 (edge clamping works on 16-B groups).  The 256x128 tile has no other staging.
bool glds_eligible(const dyn_gemm_desc* d) {
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    const bool vec = al16(d->A) && al16(d->B) && (d->lda % 4 == 0) && (d->ldb % 4 == 0) && (d->sa1 % 4 == 0) && (d->sa2 % 4 == 0) &&
                     (d->sb1 % 4 == 0) && (d->sb2 % 4 == 0);
    static const bool allow_glds = [] { const char* e = getenv("DYN_GEMM_GLDS"); return !e || atoi(e) != 0; }();
    const bool ta = d->trans_a != 0, tb = d->trans_b != 0;
    return allow_glds && vec && d->K % BK == 0 && d->K > 0 && (!ta || d->M % 4 == 0) && (tb || d->N % 4 == 0) && (!ta || d->M >= 4) && (tb || d->N >= 4);
}

// One configuration: tile (bm, bn), global split-K s, tail slicing f_req (-1 = slots / remainder, 1 = off, n = forced).
bool eval_config(const dyn_gemm_desc* d, int bm, int bn, int s, int f_req, Plan* out, double* cost) {
    const int64_t batch = d->nb1 * d->nb2;
    const int64_t ktiles = dyn::cdiv(d->K > 0 ? d->K : 1, BK);
    if (s < 1) s = 1;
    if (s > ktiles) s = (int)ktiles;
    const int64_t tiles = dyn::cdiv(d->M, bm) * dyn::cdiv(d->N, bn);
    const int64_t slots = (int64_t)tile_occ(bm, bn) * 256;
    const int64_t kchunk = dyn::cdiv(ktiles, s) * BK;
    const int s_eff = (int)dyn::cdiv(d->K > 0 ? d->K : 1, kchunk);
    const int64_t items = tiles * batch * s_eff;
    const int64_t rounds = items / slots, rem = items % slots;
    int f = 1;
    if (s_eff == 1 && rem > 0 && f_req != 1) {
        f = f_req > 1 ? f_req : (int)(slots / rem);
        const int64_t maxf = ktiles / 2 > 0 ? ktiles / 2 : 1;  // keep >= 2 K steps per slice
        if (f > maxf) f = (int)maxf;
        if (f > 16) f = 16;
        if (f < 1) f = 1;
    }
    // seconds: one round = occ tiles per CU, each 2*bm*bn*kchunk flops at eff * 614 GFLOP/s per CU
    const double round_t = tile_occ(bm, bn) * 2.0 * bm * bn * (double)kchunk / (614e9 * tile_eff(bm, bn));
    double t = rounds * round_t + (rem > 0 ? round_t / f + 1.0e-6 : 0.0);
    if (s_eff > 1) t += 3e-6 + (double)s_eff * d->M * d->N * batch * 4.0 * 2.0 / 3.0e12;
    if (f > 1) t += 2e-6 + (double)f * rem * bm * bn * 4.0 * 2.0 / 3.0e12;
    *cost = t;
    out->bm = bm; out->bn = bn; out->splits = s_eff; out->kchunk = kchunk;
    out->ws_bytes = s_eff > 1 ? (int64_t)s_eff * batch * tiles * bm * bn * (int64_t)sizeof(float) : 0;   // tile-local slabs (>= the dense layout)
    out->tail_f = f;
    out->tail_items = f > 1 ? rem : 0;
    out->full_items = items - out->tail_items;
    out->tail_ws_bytes = f > 1 ? rem * f * (int64_t)bm * bn * (int64_t)sizeof(float) : 0;
    return true;
}

struct Tuned { int ta, tb; int64_t M, N, K, batch; int bm, bn, split, tail; };
#include "gemm_tuned.inc"   // generated by scripts/tune_gemm.py on an MI355X: static const Tuned kTuned[]; kNumTuned

Plan make_plan(const dyn_gemm_desc* d) {
    Plan best;
    double cost;
    if (d->tile_m > 0 && d->tile_n > 0) {  // caller-forced configuration (autotuner, tests)
        const bool big = d->tile_m >= 256 && d->tile_n >= 128 && glds_eligible(d);       // 256x128: direct-to-LDS operands only
        eval_config(d, big ? 256 : d->tile_m >= 128 ? 128 : 64, d->tile_n >= 128 ? 128 : 64, d->split_k > 0 ? d->split_k : 1,
                    d->tail_slices > 0 ? d->tail_slices : 1, &best, &cost);
        return best;
    }
    const int64_t batch = d->nb1 * d->nb2;
    if (d->split_k == 0) {
        // the generators write the table sorted by (ta, tb, M, N, K, batch): binary search (the wav2vec2 length buckets made it ~10x longer);
        // a hand-edited, unsorted table is still found by the linear walk
        auto less = [](const Tuned& a, const Tuned& b) {
            if (a.ta != b.ta) return a.ta < b.ta;
            if (a.tb != b.tb) return a.tb < b.tb;
            if (a.M != b.M) return a.M < b.M;
            if (a.N != b.N) return a.N < b.N;
            if (a.K != b.K) return a.K < b.K;
            return a.batch < b.batch;
        };
        static const bool sorted = [&] {
            for (int i = 1; i < kNumTuned; ++i)
                if (!less(kTuned[i - 1], kTuned[i])) return false;
            return true;
        }();
        const Tuned want{d->trans_a != 0, d->trans_b != 0, d->M, d->N, d->K, batch, 0, 0, 0, 0};
        const Tuned* hit = nullptr;
        if (sorted) {
            const Tuned* it = std::lower_bound(kTuned, kTuned + kNumTuned, want, less);
            if (it != kTuned + kNumTuned && !less(want, *it)) hit = it;
        } else {
            for (int i = 0; i < kNumTuned && !hit; ++i)
                if (!less(kTuned[i], want) && !less(want, kTuned[i])) hit = &kTuned[i];
        }
        if (hit) {
            eval_config(d, (hit->bm == 256 && !glds_eligible(d)) ? 128 : hit->bm, hit->bn, hit->split, hit->tail, &best, &cost);
            return best;
        }
    }
    const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    int split_opts[9] = {1, 2, 3, 4, 6, 8, 12, 16, 24};
    int n_opts = 9;
    if (d->split_k > 0) { split_opts[0] = d->split_k; n_opts = 1; }   // caller forced a split factor
    double best_cost = 1e300;
    for (int c = 0; c < 4; ++c) {
        const int bm = cand[c][0], bn = cand[c][1];
        if ((d->M <= 64 && bm == 128) || (d->N <= 64 && bn == 128)) continue;
        for (int so = 0; so < n_opts; ++so) {
            const int s = split_opts[so];
            if (d->split_k == 0 && s > 1 && (d->K / s < 256)) continue;
            Plan pl;
            eval_config(d, bm, bn, s, d->split_k == 0 ? -1 : 1, &pl, &cost);
            if (cost < best_cost) { best_cost = cost; best = pl; }
        }
    }
    return best;
}

template <bool TA, bool TB, int BM, int BN>
void launch_vec(const KParams& kp, bool vec, dim3 grid, hipStream_t st) {
    // direct-to-LDS staging: whole K tiles only (no zero fill), 16-B aligned operands, and an operand whose rows are the
    // contiguous axis must have a row count that is a multiple of 4 (edge clamping works on 16-B groups)
    static const bool allow_glds = [] { const char* e = getenv("DYN_GEMM_GLDS"); return !e || atoi(e) != 0; }();
    const bool glds = allow_glds && vec && kp.K % BK == 0 && kp.K > 0 && (!TA || kp.M % 4 == 0) && (TB || kp.N % 4 == 0) &&
                      (!TA || kp.M >= 4) && (TB || kp.N >= 4);
    if (glds) hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, BM, BN, true, true>), grid, dim3(NTHREADS), 0, st, kp);
    else if (vec) hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, BM, BN, true, false>), grid, dim3(NTHREADS), 0, st, kp);
    else hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, BM, BN, false, false>), grid, dim3(NTHREADS), 0, st, kp);
}

template <bool TA, bool TB>
void launch_tile(const KParams& kp, const Plan& pl, bool vec, dim3 grid, hipStream_t st) {
    if (pl.bm == 256) hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, 256, 128, true, true, 8>), grid, dim3(512), 0, st, kp);   // planned only when glds_eligible()
    else if (pl.bm == 128 && pl.bn == 128) launch_vec<TA, TB, 128, 128>(kp, vec, grid, st);
    else if (pl.bm == 128 && pl.bn == 64) launch_vec<TA, TB, 128, 64>(kp, vec, grid, st);
    else if (pl.bm == 64 && pl.bn == 128) launch_vec<TA, TB, 64, 128>(kp, vec, grid, st);
    else launch_vec<TA, TB, 64, 64>(kp, vec, grid, st);
}

}  // namespace

extern "C" int64_t dyn_gemm_f32_workspace_bytes(const dyn_gemm_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->nb1 <= 0 || d->nb2 <= 0) return 0;
    const Plan pl = make_plan(d);
    return pl.ws_bytes > pl.tail_ws_bytes ? pl.ws_bytes : pl.tail_ws_bytes;
}

extern "C" int dyn_gemm_f32(const dyn_gemm_desc* d, void* stream) {
    DYN_REQUIRE(d != nullptr, DYN_E_ARG, "dyn_gemm_f32: null descriptor");
    DYN_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0 && d->nb1 >= 1 && d->nb2 >= 1, DYN_E_ARG,
                "dyn_gemm_f32: bad sizes M=%lld N=%lld K=%lld nb=%lldx%lld", (long long)d->M, (long long)d->N,
                (long long)d->K, (long long)d->nb1, (long long)d->nb2);
    if (d->M == 0 || d->N == 0) return DYN_OK;
    DYN_REQUIRE(d->A && d->B && d->C, DYN_E_ARG, "dyn_gemm_f32: null operand");
    DYN_REQUIRE(d->epilogue >= 0 && d->epilogue <= 2 && (d->epilogue != 2 || d->aux), DYN_E_ARG,
                "dyn_gemm_f32: epilogue %d needs 0 <= mode <= 2 (and aux for mode 2)", d->epilogue);
    DYN_REQUIRE(d->a_colsum == nullptr, DYN_E_ARG, "dyn_gemm_f32: a_colsum is a grouped-launch feature (dyn_gemm_f32_grouped)");
    // lda < K is allowed for a non-transposed A: overlapping rows = frames of a 1-D signal (the STFT as an implicit GEMM)
    DYN_REQUIRE(d->lda >= (d->trans_a ? d->M : 1) && d->ldb >= (d->trans_b ? d->K : 1) && d->ldc >= d->N,
                DYN_E_ARG, "dyn_gemm_f32: leading dimension smaller than the row length");
    Plan pl = make_plan(d);
    const int64_t need = pl.ws_bytes > pl.tail_ws_bytes ? pl.ws_bytes : pl.tail_ws_bytes;
    if (need > 0 && (d->workspace == nullptr || d->workspace_bytes < need)) {
        DYN_REQUIRE(d->split_k == 0, DYN_E_WORKSPACE, "dyn_gemm_f32: split_k=%d needs %lld workspace bytes, got %lld",
                    d->split_k, (long long)need, (long long)d->workspace_bytes);
        // no workspace: degrade to a single pass without K slicing (same result up to summation order)
        pl.splits = 1;
        pl.kchunk = dyn::cdiv(d->K > 0 ? d->K : 1, BK) * BK;
        pl.ws_bytes = 0; pl.tail_f = 1; pl.tail_items = 0; pl.tail_ws_bytes = 0;
        pl.full_items = dyn::cdiv(d->M, pl.bm) * dyn::cdiv(d->N, pl.bn) * d->nb1 * d->nb2;
    }
    const int64_t batch = d->nb1 * d->nb2;
    KParams kp;
    kp.M = d->M; kp.N = d->N; kp.K = d->K;
    kp.A = d->A; kp.lda = d->lda; kp.sa1 = d->sa1; kp.sa2 = d->sa2;
    kp.B = d->B; kp.ldb = d->ldb; kp.sb1 = d->sb1; kp.sb2 = d->sb2;
    kp.C = d->C; kp.ldc = d->ldc; kp.sc1 = d->sc1; kp.sc2 = d->sc2;
    kp.bias = d->bias; kp.bs1 = d->bias_s1; kp.bs2 = d->bias_s2; kp.alpha = d->alpha; kp.beta = d->beta;
    kp.cin_delta = d->C_in ? (int64_t)(d->C_in - d->C) : 0;
    kp.nb2 = d->nb2; kp.splits = pl.splits; kp.kchunk = pl.kchunk;
    kp.ws = (float*)d->workspace; kp.nbatch = batch;
    const int64_t tiles_m = dyn::cdiv(d->M, pl.bm), tiles_n = dyn::cdiv(d->N, pl.bn);
    kp.tiles_n = (int)tiles_n;
    kp.tiles_m = (int)tiles_m;
    // Each XCD (private L2) walks a contiguous run of items: row-major runs re-fetch all of B per XCD and A once overall,
    // column-major runs the opposite.  Re-fetch the SMALLER operand.
    static const bool xcd_map = [] { const char* e = getenv("DYN_GEMM_XCDMAP"); return !e || atoi(e) != 0; }();
    kp.n_major = (xcd_map && d->N > d->M) ? 1 : 0;
    kp.tiles_per_batch = tiles_m * tiles_n;
    kp.full_items = pl.full_items;
    kp.tail_f = pl.tail_f;
    kp.tail_ws = (float*)d->workspace;
    kp.epi = d->epilogue; kp.aux = d->aux;
    // in-kernel combine of K slices: needs one zeroed arrival counter per partial tile
    const int64_t n_ptiles = pl.splits > 1 ? tiles_m * tiles_n * batch : pl.tail_items;
    // The caller opts in by handing over counters.  The host wrapper does so only under DYN_GEMM_TICKET=1: measured on the 1 h job
    // (3 chains, one box, alternating runs) 577 audio-s/s with the in-kernel combine against 774 with the separate reduce pass —
    // the device-scope release / acquire fences write back and invalidate the XCD's whole L2 (buffer_wbl2 / buffer_inv sc1) under
    // every other kernel that shares it.  Kept because it is bit-identical and needs no second launch when a GEMM runs alone.
    kp.counters = (d->counters && n_ptiles > 0 && n_ptiles <= d->n_counters) ? d->counters : nullptr;
    kp.colsum = nullptr; kp.colsum_beta = 0.f; kp.first_item = 0;
    const int64_t nblocks = pl.full_items + pl.tail_items * pl.tail_f;
    DYN_REQUIRE(nblocks < (1ll << 31) && tiles_m * tiles_n < (1ll << 31), DYN_E_ARG, "dyn_gemm_f32: grid too large (%lld workgroups)",
                (long long)nblocks);
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    const bool vec = al16(d->A) && al16(d->B) && (d->lda % 4 == 0) && (d->ldb % 4 == 0) && (d->sa1 % 4 == 0) &&
                     (d->sa2 % 4 == 0) && (d->sb1 % 4 == 0) && (d->sb2 % 4 == 0);
    dim3 grid((unsigned)nblocks);
    hipStream_t st = (hipStream_t)stream;
    if (!d->trans_a && !d->trans_b) launch_tile<false, false>(kp, pl, vec, grid, st);
    else if (!d->trans_a && d->trans_b) launch_tile<false, true>(kp, pl, vec, grid, st);
    else if (d->trans_a && !d->trans_b) launch_tile<true, false>(kp, pl, vec, grid, st);
    else launch_tile<true, true>(kp, pl, vec, grid, st);
    int rc = dyn::check_launch("dyn_gemm_f32");
    if (rc != DYN_OK) return rc;
    if (kp.counters != nullptr) return rc;      // the last-arriving workgroup of every partial tile has already written C
    if (pl.splits > 1) {
        const int64_t total = d->M * d->N * batch;
        int64_t nblk = dyn::cdiv(total, 256);
        if (nblk > 2048) nblk = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, st, kp);
        rc = dyn::check_launch("dyn_gemm_f32(splitk_reduce)");
    } else if (pl.tail_items > 0) {
        int64_t nblk = dyn::cdiv(pl.tail_items * pl.bm * pl.bn / 4, 256);
        if (nblk > 4096) nblk = 4096;
        hipLaunchKernelGGL(tail_reduce_kernel, dim3((unsigned)nblk), dim3(256), 0, st, kp, pl.bm, pl.bn, pl.tail_items);
        rc = dyn::check_launch("dyn_gemm_f32(tail_reduce)");
    }
    return rc;
}

extern "C" int64_t dyn_gemm_f32_grouped_workspace_bytes(int32_t n) {
    return (int64_t)(n > 0 ? n : 0) * (int64_t)sizeof(KParams);
}

extern "C" int dyn_gemm_f32_grouped(const dyn_gemm_desc* descs, int32_t n, void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(descs != nullptr && n >= 0 && n <= kMaxGroups, DYN_E_ARG, "dyn_gemm_f32_grouped: need 0 <= n <= %d descriptors", kMaxGroups);
    if (n == 0) return DYN_OK;
    DYN_REQUIRE(workspace && workspace_bytes >= dyn_gemm_f32_grouped_workspace_bytes(n), DYN_E_WORKSPACE,
                "dyn_gemm_f32_grouped: workspace %lld < %lld bytes", (long long)workspace_bytes,
                (long long)dyn_gemm_f32_grouped_workspace_bytes(n));
    const bool ta = descs[0].trans_a != 0, tb = descs[0].trans_b != 0;
    constexpr int BM = 128, BN = 128;
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    bool glds = true;
    KParams host[kMaxGroups];
    int64_t total = 0;
    int used = 0;
    for (int g = 0; g < n; ++g) {
        const dyn_gemm_desc* d = &descs[g];
        DYN_REQUIRE((d->trans_a != 0) == ta && (d->trans_b != 0) == tb, DYN_E_ARG, "dyn_gemm_f32_grouped: group %d has other transpose flags", g);
        DYN_REQUIRE(d->nb1 == 1 && d->nb2 == 1 && d->split_k == 0 && d->epilogue == 0 && d->bias == nullptr, DYN_E_ARG,
                    "dyn_gemm_f32_grouped: group %d: batches, split-K, bias and activation epilogues are not supported here", g);
        DYN_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, DYN_E_ARG, "dyn_gemm_f32_grouped: group %d has a negative size", g);
        if (d->M == 0 || d->N == 0) continue;
        DYN_REQUIRE(d->A && d->B && d->C, DYN_E_ARG, "dyn_gemm_f32_grouped: group %d has a null operand", g);
        DYN_REQUIRE(d->lda >= (ta ? d->M : d->K) && d->ldb >= (tb ? d->K : d->N) && d->ldc >= d->N, DYN_E_ARG,
                    "dyn_gemm_f32_grouped: group %d: leading dimension smaller than the row length", g);
        DYN_REQUIRE(al16(d->A) && al16(d->B) && d->lda % 4 == 0 && d->ldb % 4 == 0, DYN_E_ARG,
                    "dyn_gemm_f32_grouped: group %d: operands must be 16-byte aligned with leading dimensions that are multiples of 4", g);
        DYN_REQUIRE(d->a_colsum == nullptr || ta, DYN_E_ARG, "dyn_gemm_f32_grouped: a_colsum needs trans_a (A stored [K][M])");
        glds = glds && d->K % BK == 0 && d->K > 0 && (!ta || (d->M % 4 == 0 && d->M >= 4)) && (tb || (d->N % 4 == 0 && d->N >= 4));
        KParams& kp = host[used++];
        kp.M = d->M; kp.N = d->N; kp.K = d->K;
        kp.A = d->A; kp.lda = d->lda; kp.sa1 = 0; kp.sa2 = 0;
        kp.B = d->B; kp.ldb = d->ldb; kp.sb1 = 0; kp.sb2 = 0;
        kp.C = d->C; kp.ldc = d->ldc; kp.sc1 = 0; kp.sc2 = 0;
        kp.cin_delta = d->C_in ? (int64_t)(d->C_in - d->C) : 0;
        kp.bias = nullptr; kp.bs1 = 0; kp.bs2 = 0; kp.alpha = d->alpha; kp.beta = d->beta;
        kp.nb2 = 1; kp.splits = 1; kp.kchunk = dyn::cdiv(d->K > 0 ? d->K : 1, BK) * BK;
        kp.ws = nullptr; kp.nbatch = 1;
        kp.tiles_m = (int)dyn::cdiv(d->M, BM); kp.tiles_n = (int)dyn::cdiv(d->N, BN);
        kp.n_major = 0;       // row-major runs of tiles: consecutive tiles share the A panel (dy), the operand the bias sums also read
        kp.tiles_per_batch = (int64_t)kp.tiles_m * kp.tiles_n;
        kp.full_items = kp.tiles_per_batch; kp.tail_f = 1; kp.tail_ws = nullptr;
        kp.epi = 0; kp.aux = nullptr; kp.counters = nullptr;
        kp.colsum = d->a_colsum; kp.colsum_beta = d->a_colsum_beta;
        kp.first_item = total;
        total += kp.tiles_per_batch;
    }
    if (used == 0) return DYN_OK;
    DYN_REQUIRE(total < (1ll << 31), DYN_E_ARG, "dyn_gemm_f32_grouped: grid too large");
    static const bool allow_glds = [] { const char* e = getenv("DYN_GEMM_GLDS"); return !e || atoi(e) != 0; }();
    glds = glds && allow_glds;
    hipStream_t st = (hipStream_t)stream;
    KParams* table = (KParams*)workspace;
    for (int base = 0; base < used; base += kTableChunk) {
        KTableChunk chunk;
        const int cnt = used - base < kTableChunk ? used - base : kTableChunk;
        for (int t = 0; t < cnt; ++t) chunk.g[t] = host[base + t];
        for (int t = cnt; t < kTableChunk; ++t) chunk.g[t] = host[base];
        hipLaunchKernelGGL(gemm_table_kernel, dim3(1), dim3(64), 0, st, chunk, table, base, cnt);
    }
    dim3 grid((unsigned)total);
#define GO_G(TA_, TB_)                                                                                                              \
    do {                                                                                                                            \
        if (glds) hipLaunchKernelGGL((gemm_f32_grouped_kernel<TA_, TB_, BM, BN, true>), grid, dim3(NTHREADS), 0, st, table, used, total); \
        else hipLaunchKernelGGL((gemm_f32_grouped_kernel<TA_, TB_, BM, BN, false>), grid, dim3(NTHREADS), 0, st, table, used, total);     \
    } while (0)
    if (ta && !tb) GO_G(true, false);
    else if (!ta && tb) GO_G(false, true);
    else if (!ta && !tb) GO_G(false, false);
    else GO_G(true, true);
#undef GO_G
    return dyn::check_launch("dyn_gemm_f32_grouped");
}
