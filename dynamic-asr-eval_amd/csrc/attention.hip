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
                                                            int64_t out_batch_stride, float scale) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TILE];   // 2 stages x (K tile, V tile) = 64 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int64_t b = blockIdx.z, head = blockIdx.y;
    const int64_t q0 = (int64_t)blockIdx.x * BQ + wave * 32;
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

    const int64_t ntiles = (T + BKEY - 1) / BKEY;
    set_src(0);
    issue(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int64_t kt = 0; kt < ntiles; ++kt) {
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
}  // namespace

// q / k / v: [B, T, .] views with row stride `row_stride` and batch stride `batch_stride` (floats), head h at +h * 128 (so the
// packed [B, T, 3 * H * 128] QKV activation is passed as three base pointers); out [B, T, H * 128]-like with its own strides.
extern "C" int dyn_attention_fwd(const float* q, const float* k, const float* v, float* out, int64_t B, int64_t T, int64_t H,
                                 int64_t head_dim, int64_t row_stride, int64_t batch_stride, int64_t out_row_stride,
                                 int64_t out_batch_stride, float scale, void* stream) {
    DYN_REQUIRE(q && k && v && out && B >= 0 && T >= 0 && H > 0, DYN_E_ARG, "dyn_attention_fwd: bad arguments");
    DYN_REQUIRE(head_dim == D, DYN_E_UNSUPPORTED, "dyn_attention_fwd: head_dim %lld (the fused kernel is built for 128)", (long long)head_dim);
    DYN_REQUIRE(row_stride % 4 == 0 && batch_stride % 4 == 0 && out_row_stride % 4 == 0 && out_batch_stride % 4 == 0 &&
                    ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) == 0, DYN_E_ARG,
                "dyn_attention_fwd: q / k / v / out must be 16-byte aligned with strides that are multiples of 4 floats");
    if (B == 0 || T == 0) return DYN_OK;
    dim3 grid((unsigned)dyn::cdiv(T, BQ), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL(attention_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, out, T, row_stride, batch_stride,
                       out_row_stride, out_batch_stride, scale);
    return dyn::check_launch("dyn_attention_fwd");
}
