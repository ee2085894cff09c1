// Fused self-attention forward for the no-grad passes (final pass of the dynamic-eval loop, epochs = 0 baselines; reference
// lcasr/lib.py:603 runs model(audio_signal) under torch.no_grad()): O = softmax(Q K^T * scale) V per (batch, head), fp32 on the
// matrix cores (v_mfma_f32_32x32x2_f32), K/V tiles staged in LDS by direct-to-LDS loads, online softmax — the [T, T] score matrix
// never reaches HBM (201 MB per block at B = 2, T' = 2048 in the unfused path).  The grad-mode forward keeps the unfused
// GEMM -> softmax -> GEMM sequence because its backward needs the probabilities.
//
// One 256-thread workgroup = 128 query rows of one (b, h); wave w owns rows 32w .. 32w+31.  Per 32-key tile:
//   S^T = K_tile Q^T  (A = K from LDS, B = Q held in registers for the whole kernel)   64 MFMAs
//   online softmax ALONG REGISTERS: in the MFMA C layout a lane holds one query column and 16 keys, so the row maximum / sum are
//   in-lane reductions plus one exchange between the two lane halves;
//   O += P V: the C layout of S^T (lane = query, register e = key 8c + 4h + q', e = 4c + q') IS the A-operand layout of the
//   second product in this kernel's k order, so P feeds the matrix core straight from the accumulator registers — no transpose,
//   no LDS round trip.                                                                                               64 MFMAs
// K tile: [key][128] with the 16-B slot q of key r stored at slot q ^ (r & 15) (conflict-free ds_read_b128 across 16 keys; the
// swizzle is applied to the SOURCE address of the direct-to-LDS load); V tile: [key][128] linear, one ds_read_b128 per key.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

namespace {
constexpr int D = 128, BQ = 128, BKEY = 32, TILE = BKEY * D;   // floats per K or V tile

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ out, int64_t T,
                                                            int64_t row_stride, int64_t batch_stride, int64_t out_row_stride,
                                                            int64_t out_batch_stride, float scale, float* __restrict__ lse,
                                                            int nsplit, int64_t out_split_stride, int64_t lse_split_stride) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TILE];   // 2 stages x (K tile, V tile) = 64 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int64_t b = blockIdx.z, head = blockIdx.y;
    // key split (nsplit > 1): workgroup (query block, split) walks only its share of the key tiles and writes a NORMALISED partial
    // output plus its log-sum-exp to the split's slab; attention_merge_kernel combines the slabs (the launch then has nsplit times
    // the workgroups: 4 x 6 x 16 = 384 workgroups of the B = 4 final pass fill 0.75 of the 512 slots, 768 half-size ones balance)
    const int split = nsplit > 1 ? (int)(blockIdx.x % nsplit) : 0;
    const int64_t qblock = nsplit > 1 ? blockIdx.x / nsplit : blockIdx.x;
    const int64_t q0 = qblock * BQ + wave * 32;
    out += (int64_t)split * out_split_stride;
    if (lse != nullptr) lse += (int64_t)split * lse_split_stride;
    const float* qb = q + b * batch_stride + head * D;
    const float* kb = k + b * batch_stride + head * D;
    const float* vb = v + b * batch_stride + head * D;

    // this lane's query row (clamped at the edge; such rows are never stored), scaled once: 16 chunks of (8c + 4h .. +3)
    const int64_t qrow = (q0 + i < T) ? q0 + i : T - 1;
    float qf[16][4];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(qb + qrow * row_stride + 8 * c + 4 * h);
        qf[c][0] = t.x * scale; qf[c][1] = t.y * scale; qf[c][2] = t.z * scale; qf[c][3] = t.w * scale;
    }

    // direct-to-LDS pieces: a K or V tile is 16 KiB = 16 pieces of 1 KiB (2 key rows each); wave w issues pieces 4w .. 4w+3
    const float* ksrc[4];
    const float* vsrc[4];
    const int prow_in = lane >> 5, pslot = lane & 31;
    auto set_src = [&](int64_t key0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (wave * 4 + j) * 2 + prow_in;              // key row inside the tile
            int64_t key = key0 + r;
            key = key < T ? key : T - 1;                               // masked later; keeps the address valid and the data finite
            ksrc[j] = kb + key * row_stride + 4 * (pslot ^ (r & 15));
            vsrc[j] = vb + key * row_stride + 4 * pslot;
        }
    };
    // The loads are issued from inline assembly: with the builtin, hipcc (ROCm 7.2) waits vmcnt(0) at the first LDS read after an
    // LDS-DMA it cannot prove disjoint — here the top of every iteration, which exposed the whole load latency.  The tile in flight
    // is waited for explicitly before the barrier that ends the iteration (M0 = LDS destination, saved and restored around the load).
    auto glds16 = [&](const float* src, float* dst) {
        unsigned keep;
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)dst);   // wave-uniform: an SGPR
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
    };
    auto issue = [&](float* stage) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(ksrc[j], stage + (wave * 4 + j) * 256);
            glds16(vsrc[j], stage + TILE + (wave * 4 + j) * 256);
        }
    };

    f32x16 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;     // running maximum / sum of this lane's query (both lane halves hold the same values)

    const int64_t ntiles_all = (T + BKEY - 1) / BKEY;
    const int64_t per_split = (ntiles_all + nsplit - 1) / nsplit;
    const int64_t kt0 = split * per_split;
    const int64_t ntiles = kt0 + per_split < ntiles_all ? kt0 + per_split : ntiles_all;     // the host guarantees kt0 < ntiles
    set_src(kt0 * BKEY);
    issue(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int64_t kt = kt0; kt < ntiles; ++kt) {
        const float* ks = smem + cur * 2 * TILE;
        const float* vs = ks + TILE;
        if (kt + 1 < ntiles) {
            set_src((kt + 1) * BKEY);
            issue(smem + (cur ^ 1) * 2 * TILE);
        }
        // S^T tile: rows = keys (A = K from LDS), columns = this wave's 32 queries (B = qf)
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 kf = *reinterpret_cast<const float4*>(&ks[i * D + 4 * ((2 * c + h) ^ (i & 15))]);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[c][0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[c][1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[c][2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[c][3], s, 0, 0, 0);
        }
        // register e of lane (query i, half h) = key kt*32 + (e & 3) + 8 * (e >> 2) + 4 * h
        const int64_t key_base = kt * BKEY + 4 * h;
        float mt = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (key_base + (e & 3) + 8 * (e >> 2) >= T) s[e] = -INFINITY;
            mt = fmaxf(mt, s[e]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);        // first tile: exp(-inf) = 0
        float lt = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = __expf(s[e] - m_new); lt += s[e]; }
        lt += __shfl_xor(lt, 32, 64);
        l_run = l_run * alpha + lt;
        m_run = m_new;
        // O rows are queries indexed by (register, half); alpha lives in the lane of its query: broadcast it
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float a = __shfl(alpha, (e & 3) + 8 * (e >> 2) + 4 * h, 64);
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t][e] *= a;
        }
        // O += P V: A = P straight from the accumulator registers (e = 4c + q'), B = V[key 8c + 4h + q'][4i + t]: output tile t
        // holds the head-dim columns 4i + t, so ONE ds_read_b128 per key feeds all four tiles (and the final store is 16 B per lane)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const float4 vv = *reinterpret_cast<const float4*>(vs + (8 * c + 4 * h + qq) * D + 4 * i);
                o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[4 * c + qq], vv.x, o[0], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[4 * c + qq], vv.y, o[1], 0, 0, 0);
                o[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[4 * c + qq], vv.z, o[2], 0, 0, 0);
                o[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[4 * c + qq], vv.w, o[3], 0, 0, 0);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the next tile has landed ...
        __syncthreads();                                      // ... everybody's has, and every wave is done reading this stage
        cur ^= 1;
    }
    // grad mode: the backward re-forms P = exp(s - lse) tile by tile from this one number per query row
    if (lse != nullptr && h == 0 && q0 + i < T) lse[(b * gridDim.y + head) * T + q0 + i] = m_run + __logf(l_run);
    // normalise and store: O[query (e, h)][4 i + t]
    const float inv = 1.f / l_run;
    float* ob = out + b * out_batch_stride + head * D;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * h;
        const float w = __shfl(inv, r, 64);
        const int64_t row = q0 + r;
        if (row < T)
            *reinterpret_cast<float4*>(ob + row * out_row_stride + 4 * i) = make_float4(o[0][e] * w, o[1][e] * w, o[2][e] * w, o[3][e] * w);
    }
}

// out[b, row, head, :] = sum_s w_s * part_s[b, row, head, :],  w_s = exp(lse_s - lse),  lse = log sum_s exp(lse_s)  (fixed order in s).
// One thread per 16 B of output; part / out share the row and batch strides, the slabs are `split_stride` floats apart.
__global__ __launch_bounds__(256) void attention_merge_kernel(const float* __restrict__ part, const float* __restrict__ part_lse,
                                                              float* __restrict__ out, float* __restrict__ lse, int64_t B, int64_t T, int64_t H,
                                                              int nsplit, int64_t part_row_stride, int64_t part_batch_stride,
                                                              int64_t part_split_stride, int64_t lse_split_stride, int64_t out_row_stride,
                                                              int64_t out_batch_stride) {
    const int64_t total = B * T * H * (D / 4);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % (D / 4));
        const int64_t r = idx / (D / 4);
        const int64_t head = r % H, row = (r / H) % T, b = r / (H * T);
        const float* lp = part_lse + (b * H + head) * T + row;
        float m = -INFINITY;
        for (int s = 0; s < nsplit; ++s) m = fmaxf(m, lp[(int64_t)s * lse_split_stride]);
        float den = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < nsplit; ++s) {
            const float w = __expf(lp[(int64_t)s * lse_split_stride] - m);
            const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)s * part_split_stride + b * part_batch_stride + row * part_row_stride + head * D + 4 * c4);
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
            den += w;
        }
        const float inv = 1.f / den;
        *reinterpret_cast<float4*>(out + b * out_batch_stride + row * out_row_stride + head * D + 4 * c4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
        if (lse != nullptr && c4 == 0) lse[(b * H + head) * T + row] = m + __logf(den);
    }
}

// Key splits of a launch.  Measured at T' = 2048, H = 6 (profiles/r03_probe_attention.log): the kernel is fastest when the launch has
// 768 workgroups = exactly three per CU (two resident, the third follows): B = 4 (384 query-block workgroups) 574 us unsplit, 443 us
// with 2 splits, 495 with 3 (2.25 per CU: unbalanced), 454 with 4; B = 2 (192): 304 -> 234 us with 4; B = 1 (96): 299 -> 130 us with 8.
// So: the split count that brings the launch closest to 768 workgroups, at least 8 key tiles (256 keys) per split, at most 8 splits.
int pick_splits(int64_t B, int64_t T, int64_t H) {
    const int64_t wgs = B * H * dyn::cdiv(T, BQ);
    const int64_t ntiles = dyn::cdiv(T, BKEY);
    if (wgs >= 640 || ntiles < 16) return 1;
    int64_t s = (768 + wgs / 2) / wgs;
    if (s > ntiles / 8) s = ntiles / 8;
    if (s > 8) s = 8;
    return s < 1 ? 1 : (int)s;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward of the fused attention (grad-mode windows of Loop A, reference lcasr/lib.py:550,579), recomputing the probabilities tile
// by tile from (Q, K, lse) instead of reading a materialised [T, T] matrix.  Deterministic: no float atomics —
//   KEY_OWNER = false: one workgroup owns 128 QUERY rows, walks all key tiles:  dQ = scale * sum_j dS_ij K_j          (3 products)
//                      and writes delta_i = sum_d dO_i O_i for the second kernel;
//   KEY_OWNER = true : one workgroup owns 128 KEY rows, walks all query tiles:  dV = sum_i P_ij dO_i,  dK = scale * sum_i dS_ij Q_i
//                                                                                                                       (4 products)
// with dS = P o (dP - delta), dP = dO V^T, P = exp(scale * Q K^T - lse).  Both are the forward kernel's structure: the owner's rows
// live in registers as MFMA B operands, the other side arrives in 32-row tiles by direct-to-LDS loads (two tensors, two stages,
// 64 KB), s / dp come out in the C layout "lane = owner row, register = tile row", and P^T / dS^T in that layout ARE the A operands
// of the accumulating products (no transpose, no LDS round trip).  Both staged tensors are stored with the source-address swizzle
// slot ^ (row & 15): conflict-free for the A-operand reads (lane = row) and for the B-operand reads (lane = 16-B column slot).
template <bool KEY_OWNER>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            const float* __restrict__ o, const float* __restrict__ dout,
                                                            const float* __restrict__ lse, float* __restrict__ delta,
                                                            float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                            int64_t T, int64_t row_stride, int64_t batch_stride, int64_t o_row_stride,
                                                            int64_t o_batch_stride, int64_t g_row_stride, int64_t g_batch_stride, float scale) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TILE];
    __shared__ __attribute__((aligned(16))) float rowstat[2][2][BKEY];     // [stage][lse | delta][tile row]   (KEY_OWNER only)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int64_t b = blockIdx.z, head = blockIdx.y, H = gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.x * BQ + wave * 32;                 // first owner row of this wave
    const float* qb = q + b * batch_stride + head * D;
    const float* kb = k + b * batch_stride + head * D;
    const float* vb = v + b * batch_stride + head * D;
    const float* ob = o + b * o_batch_stride + head * D;
    const float* gb = dout + b * o_batch_stride + head * D;
    const float* lse_b = lse + (b * H + head) * T;
    float* delta_b = delta + (b * H + head) * T;

    // owner rows in registers (B-operand layout: lane (i, h) holds row i, d = 8c + 4h .. +3)
    const int64_t orow = (r0 + i < T) ? r0 + i : T - 1;
    float f1[16][4], f2[16][4];              // query owner: scale * Q, dO          key owner: scale * K, V
    float my_lse = 0.f, my_delta = 0.f;
    {
        const float* s1 = KEY_OWNER ? kb : qb;
        const float* s2 = KEY_OWNER ? vb : gb;
        const int64_t st2 = KEY_OWNER ? row_stride : o_row_stride;
        float dsum = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(s1 + orow * row_stride + 8 * c + 4 * h);
            const float4 g = *reinterpret_cast<const float4*>(s2 + orow * st2 + 8 * c + 4 * h);
            f1[c][0] = a.x * scale; f1[c][1] = a.y * scale; f1[c][2] = a.z * scale; f1[c][3] = a.w * scale;
            f2[c][0] = g.x; f2[c][1] = g.y; f2[c][2] = g.z; f2[c][3] = g.w;
            if (!KEY_OWNER) {
                const float4 oo = *reinterpret_cast<const float4*>(ob + orow * o_row_stride + 8 * c + 4 * h);
                dsum += (g.x * oo.x + g.y * oo.y) + (g.z * oo.z + g.w * oo.w);
            }
        }
        if (!KEY_OWNER) {
            my_delta = dsum + __shfl_xor(dsum, 32, 64);
            my_lse = lse_b[orow];
            if (h == 0 && r0 + i < T) delta_b[r0 + i] = my_delta;
        }
    }

    // tiles of the other side: query owner walks (K, V); key owner walks (Q, dO)
    const float* t1b = KEY_OWNER ? qb : kb;
    const float* t2b = KEY_OWNER ? gb : vb;
    const int64_t t1s = row_stride, t2s = KEY_OWNER ? o_row_stride : row_stride;
    // source addresses are recomputed per tile from the tile's first row (a few VALU ops per 192+ MFMAs) instead of living in 16 VGPRs
    const int prow_in = lane >> 5, pslot = lane & 31;
    auto glds16 = [&](const float* src, float* dst) {
        unsigned keep;
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)dst);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
    };
    auto issue = [&](int stage, int64_t row0) {
        float* st = smem + stage * 2 * TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (wave * 4 + j) * 2 + prow_in;
            int64_t row = row0 + r;
            row = row < T ? row : T - 1;
            const int sw = 4 * (pslot ^ (r & 15));
            glds16(t1b + row * t1s + sw, st + (wave * 4 + j) * 256);
            glds16(t2b + row * t2s + sw, st + TILE + (wave * 4 + j) * 256);
        }
        if (KEY_OWNER && threadIdx.x < 2 * BKEY) {       // lse / delta of the tile's 32 query rows
            const int which = threadIdx.x >> 5, r = threadIdx.x & 31;
            const int64_t row = row0 + r < T ? row0 + r : T - 1;
            rowstat[stage][which][r] = which == 0 ? lse_b[row] : delta_b[row];
        }
    };

    f32x16 acc1[4], acc2[4];                 // query owner: acc1 = dQ          key owner: acc1 = dK, acc2 = dV
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc1[t][e] = 0.f; acc2[t][e] = 0.f; }

    const int64_t ntiles = (T + BKEY - 1) / BKEY;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int64_t kt = 0; kt < ntiles; ++kt) {
        const float* t1 = smem + cur * 2 * TILE;
        const float* t2 = t1 + TILE;
        if (kt + 1 < ntiles) issue(cur ^ 1, (kt + 1) * BKEY);
        // s[tile row][owner row] and dp[tile row][owner row]: A = tile (rows from LDS), B = owner registers
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 a1 = *reinterpret_cast<const float4*>(&t1[i * D + 4 * ((2 * c + h) ^ (i & 15))]);
            const float4 a2 = *reinterpret_cast<const float4*>(&t2[i * D + 4 * ((2 * c + h) ^ (i & 15))]);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, f1[c][0], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, f2[c][0], dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, f1[c][1], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, f2[c][1], dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, f1[c][2], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.z, f2[c][2], dp, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, f1[c][3], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.w, f2[c][3], dp, 0, 0, 0);
        }
        // register e of lane (owner i, half h) = tile row (e & 3) + 8 * (e >> 2) + 4 * h.  p = exp(s - lse[query]), ds = p * (dp - delta[query]);
        // tile rows past T contribute nothing (their loads were clamped to a valid row).
        const int64_t row_base = kt * BKEY + 4 * h;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float l4[4], d4[4];
            if (KEY_OWNER) {
                const float4 lv = *reinterpret_cast<const float4*>(&rowstat[cur][0][8 * c + 4 * h]);
                const float4 dv4 = *reinterpret_cast<const float4*>(&rowstat[cur][1][8 * c + 4 * h]);
                l4[0] = lv.x; l4[1] = lv.y; l4[2] = lv.z; l4[3] = lv.w;
                d4[0] = dv4.x; d4[1] = dv4.y; d4[2] = dv4.z; d4[3] = dv4.w;
            }
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int e = 4 * c + qq;
                const bool live = row_base + 8 * c + qq < T;
                const float pe = live ? __expf(s[e] - (KEY_OWNER ? l4[qq] : my_lse)) : 0.f;
                s[e] = pe;                                                        // P^T (query owner: P^T[key][query]; key owner: P[query][key])
                dp[e] = pe * (dp[e] - (KEY_OWNER ? d4[qq] : my_delta));          // dS in the same layout
            }
        }
        // accumulate: A = s / dp straight from the accumulator registers (k = tile row 8c + 4h + qq), B = tile[row][4i + t]
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int r = 8 * c + 4 * h + qq;
                const float4 b1 = *reinterpret_cast<const float4*>(t1 + r * D + 4 * (i ^ (r & 15)));
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[4 * c + qq], b1.x, acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[4 * c + qq], b1.y, acc1[1], 0, 0, 0);
                acc1[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[4 * c + qq], b1.z, acc1[2], 0, 0, 0);
                acc1[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[4 * c + qq], b1.w, acc1[3], 0, 0, 0);
                if (KEY_OWNER) {
                    const float4 b2 = *reinterpret_cast<const float4*>(t2 + r * D + 4 * (i ^ (r & 15)));
                    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[4 * c + qq], b2.x, acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[4 * c + qq], b2.y, acc2[1], 0, 0, 0);
                    acc2[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[4 * c + qq], b2.z, acc2[2], 0, 0, 0);
                    acc2[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[4 * c + qq], b2.w, acc2[3], 0, 0, 0);
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    // store: rows = owner rows (register, half), columns 4i + t.  The tile-side operand of acc1 (K or Q) was NOT pre-scaled: apply scale.
    float* out1 = (KEY_OWNER ? dk : dq) + b * g_batch_stride + head * D;
    float* out2 = dv + b * g_batch_stride + head * D;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int64_t row = r0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < T) {
            *reinterpret_cast<float4*>(out1 + row * g_row_stride + 4 * i) =
                make_float4(acc1[0][e] * scale, acc1[1][e] * scale, acc1[2][e] * scale, acc1[3][e] * scale);
            if (KEY_OWNER)
                *reinterpret_cast<float4*>(out2 + row * g_row_stride + 4 * i) = make_float4(acc2[0][e], acc2[1][e], acc2[2][e], acc2[3][e]);
        }
    }
}
}  // namespace

// q / k / v: [B, T, .] views with row stride `row_stride` and batch stride `batch_stride` (floats), head h at +h * 128 (so the
// packed [B, T, 3 * H * 128] QKV activation is passed as three base pointers); out [B, T, H * 128]-like with its own strides.
namespace {
int64_t split_ws_bytes(int64_t B, int64_t T, int64_t H, int nsplit) {
    return nsplit > 1 ? (int64_t)nsplit * (B * T * H * D + B * H * T) * (int64_t)sizeof(float) : 0;
}

int launch_fwd(const char* what, const float* q, const float* k, const float* v, float* out, float* lse, int64_t B, int64_t T, int64_t H,
               int64_t head_dim, int64_t row_stride, int64_t batch_stride, int64_t out_row_stride, int64_t out_batch_stride, float scale,
               int nsplit, void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(q && k && v && out && B >= 0 && T >= 0 && H > 0, DYN_E_ARG, "%s: bad arguments", what);
    DYN_REQUIRE(head_dim == D, DYN_E_UNSUPPORTED, "%s: head_dim %lld (the fused kernel is built for 128)", what, (long long)head_dim);
    DYN_REQUIRE(row_stride % 4 == 0 && batch_stride % 4 == 0 && out_row_stride % 4 == 0 && out_batch_stride % 4 == 0 &&
                    ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) == 0, DYN_E_ARG,
                "%s: q / k / v / out must be 16-byte aligned with strides that are multiples of 4 floats", what);
    if (B == 0 || T == 0) return DYN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (nsplit == 0) nsplit = workspace ? pick_splits(B, T, H) : 1;
    const int64_t ntiles = dyn::cdiv(T, BKEY);
    DYN_REQUIRE(nsplit >= 1 && nsplit <= 8 && (nsplit == 1 || (int64_t)(nsplit - 1) * dyn::cdiv(ntiles, nsplit) < ntiles), DYN_E_ARG,
                "%s: %d key splits do not fit %lld key tiles", what, nsplit, (long long)ntiles);
    if (nsplit == 1) {
        dim3 grid((unsigned)dyn::cdiv(T, BQ), (unsigned)H, (unsigned)B);
        hipLaunchKernelGGL(attention_fwd_kernel, grid, dim3(256), 0, st, q, k, v, out, T, row_stride, batch_stride, out_row_stride,
                           out_batch_stride, scale, lse, 1, (int64_t)0, (int64_t)0);
        return dyn::check_launch(what);
    }
    DYN_REQUIRE(workspace && workspace_bytes >= split_ws_bytes(B, T, H, nsplit) && (((uintptr_t)workspace) & 15) == 0, DYN_E_WORKSPACE,
                "%s: %d key splits need %lld workspace bytes (16-byte aligned), got %lld", what, nsplit,
                (long long)split_ws_bytes(B, T, H, nsplit), (long long)workspace_bytes);
    float* part = (float*)workspace;                       // [nsplit][B][T][H * D]
    const int64_t part_split = B * T * H * D;
    float* part_lse = part + (int64_t)nsplit * part_split;   // [nsplit][B][H][T]
    const int64_t lse_split = B * H * T;
    dim3 grid((unsigned)(dyn::cdiv(T, BQ) * nsplit), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL(attention_fwd_kernel, grid, dim3(256), 0, st, q, k, v, part, T, row_stride, batch_stride, H * D, T * H * D, scale,
                       part_lse, nsplit, part_split, lse_split);
    int64_t g = dyn::cdiv(B * T * H * (D / 4), 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(attention_merge_kernel, dim3((unsigned)g), dim3(256), 0, st, part, part_lse, out, lse, B, T, H, nsplit, H * D, T * H * D,
                       part_split, lse_split, out_row_stride, out_batch_stride);
    return dyn::check_launch(what);
}
}  // namespace

extern "C" int dyn_attention_fwd(const float* q, const float* k, const float* v, float* out, int64_t B, int64_t T, int64_t H,
                                 int64_t head_dim, int64_t row_stride, int64_t batch_stride, int64_t out_row_stride,
                                 int64_t out_batch_stride, float scale, void* stream) {
    return launch_fwd("dyn_attention_fwd", q, k, v, out, nullptr, B, T, H, head_dim, row_stride, batch_stride, out_row_stride, out_batch_stride,
                      scale, 1, nullptr, 0, stream);
}

extern "C" int dyn_attention_fwd_lse(const float* q, const float* k, const float* v, float* out, float* lse, int64_t B, int64_t T,
                                     int64_t H, int64_t head_dim, int64_t row_stride, int64_t batch_stride, int64_t out_row_stride,
                                     int64_t out_batch_stride, float scale, void* stream) {
    DYN_REQUIRE(lse, DYN_E_ARG, "dyn_attention_fwd_lse: lse is NULL");
    return launch_fwd("dyn_attention_fwd_lse", q, k, v, out, lse, B, T, H, head_dim, row_stride, batch_stride, out_row_stride, out_batch_stride,
                      scale, 1, nullptr, 0, stream);
}

extern "C" int64_t dyn_attention_fwd_split_workspace_bytes(int64_t B, int64_t T, int64_t H, int32_t nsplit) {
    if (B <= 0 || T <= 0 || H <= 0) return 0;
    return split_ws_bytes(B, T, H, nsplit == 0 ? pick_splits(B, T, H) : nsplit);
}

extern "C" int dyn_attention_fwd_split(const float* q, const float* k, const float* v, float* out, float* lse, int64_t B, int64_t T,
                                       int64_t H, int64_t head_dim, int64_t row_stride, int64_t batch_stride, int64_t out_row_stride,
                                       int64_t out_batch_stride, float scale, int32_t nsplit, void* workspace, int64_t workspace_bytes,
                                       void* stream) {
    return launch_fwd("dyn_attention_fwd_split", q, k, v, out, lse, B, T, H, head_dim, row_stride, batch_stride, out_row_stride,
                      out_batch_stride, scale, nsplit, workspace, workspace_bytes, stream);
}

extern "C" int dyn_attention_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout, const float* lse,
                                 float* delta, float* dq, float* dk, float* dv, int64_t B, int64_t T, int64_t H, int64_t head_dim,
                                 int64_t row_stride, int64_t batch_stride, int64_t out_row_stride, int64_t out_batch_stride,
                                 int64_t grad_row_stride, int64_t grad_batch_stride, float scale, void* stream) {
    DYN_REQUIRE(q && k && v && out && dout && lse && delta && dq && dk && dv && B >= 0 && T >= 0 && H > 0, DYN_E_ARG,
                "dyn_attention_bwd: bad arguments");
    DYN_REQUIRE(head_dim == D, DYN_E_UNSUPPORTED, "dyn_attention_bwd: head_dim %lld (the fused kernel is built for 128)", (long long)head_dim);
    DYN_REQUIRE(row_stride % 4 == 0 && batch_stride % 4 == 0 && out_row_stride % 4 == 0 && out_batch_stride % 4 == 0 &&
                    grad_row_stride % 4 == 0 && grad_batch_stride % 4 == 0 &&
                    ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out) | ((uintptr_t)dout) | ((uintptr_t)dq) |
                      ((uintptr_t)dk) | ((uintptr_t)dv)) & 15) == 0, DYN_E_ARG,
                "dyn_attention_bwd: tensors must be 16-byte aligned with strides that are multiples of 4 floats");
    if (B == 0 || T == 0) return DYN_OK;
    dim3 grid((unsigned)dyn::cdiv(T, BQ), (unsigned)H, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    // query owners first: they also produce delta, which the key owners read
    hipLaunchKernelGGL((attention_bwd_kernel<false>), grid, dim3(256), 0, st, q, k, v, out, dout, lse, delta, dq, dk, dv, T, row_stride,
                       batch_stride, out_row_stride, out_batch_stride, grad_row_stride, grad_batch_stride, scale);
    hipLaunchKernelGGL((attention_bwd_kernel<true>), grid, dim3(256), 0, st, q, k, v, out, dout, lse, delta, dq, dk, dv, T, row_stride,
                       batch_stride, out_row_stride, out_batch_stride, grad_row_stride, grad_batch_stride, scale);
    return dyn::check_launch("dyn_attention_bwd");
}
