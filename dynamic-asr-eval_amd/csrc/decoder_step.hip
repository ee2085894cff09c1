// One-token decoder steps of the autoregressive decode (`model.generate`, reference call sites lcasr/lib.py:1128,1579-1582,1620-1625):
// every product of a step is a [1, K] row times a [N, K] weight, 0.13 - 2 MFLOP, and the whole decoder (2 x 256 by default) is
// ~11 MB of weights that stay in L2 — the step is bound by launches, not by bytes or flops.  So a step is 8 * layers + 2 launches of
// two lean kernels, queued back to back from ONE C call for several tokens:
//   dec_gemv_kernel   y = act(W . f(x) + b) (+ residual): a wave per output row (coalesced 16-byte weight reads, one wave reduction per
//                     row), x held in registers; f = LayerNorm (statistics recomputed per wave from the 1 - 8 KB row: cheaper than a
//                     launch) and, for the first layer, the token-embedding + position gather itself
//   dec_attn_kernel   one query against n cached keys / values, four workgroups per head (a quarter of the keys each): scores into LDS
//                     (16 lanes per key row, four rows in flight), exponentials, weighted sum of the value rows; the four (sum, max,
//                     denominator) triples are merged in the prologue of the projection that consumes them
//   dec_pick_kernel   next token = argmax (or the Gumbel-max draw) over the logits, written straight into the token buffer
// No tile kernel, no split-K reduce, no intermediate [1, n] score matrices in HBM.
#include "common.h"

namespace {

struct GemvArgs {
    const float* x;        // [K] input row (unused with EMBED)
    const int32_t* tok;    // EMBED: the token id of this position
    const float* table;    // EMBED: [vocab, K]
    const float* pos;      // EMBED: positional row [K]
    float* x_out;          // EMBED: the gathered row is written here once (the residual stream)
    int vocab;
    const float* parts;    // MERGE: the key splits of the attention, [heads][NSPLIT][hd + 4] (unnormalised sum | max | denominator)
    int hd;
    const float* gamma;    // LN
    const float* beta;
    float eps;
    const float* W;        // [N, K]
    const float* bias;     // [N] or null
    const float* res;      // [N] or null; may alias y (element n is read and written by the same lane)
    float* y;              // [N]
    int N, rpw;            // rows per wave
};

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

constexpr int NSPLIT = 4;   // key splits of the one-query attention (one workgroup each); merged in the consumer's prologue

template <int NV, bool LN, bool SILU, bool EMBED, bool MERGE = false>
__global__ __launch_bounds__(256) void dec_gemv_kernel(const GemvArgs a) {
    constexpr int K = NV * 256;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 xv[NV];
    if (EMBED) {
        int64_t id = a.tok[0];
        id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);       // ids are validated on the host; never read outside the table
        const float4* row = reinterpret_cast<const float4*>(a.table + id * K);
        const float4* pr = reinterpret_cast<const float4*>(a.pos);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 e = row[lane + 64 * j], p = pr[lane + 64 * j];
            xv[j] = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
        }
        if (blockIdx.x == 0 && w == 0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) reinterpret_cast<float4*>(a.x_out)[lane + 64 * j] = xv[j];
        }
    } else if (MERGE) {     // x = softmax-weighted value sum of one head, put together from its NSPLIT key splits
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = 4 * (lane + 64 * j), h = c / a.hd, cc = c % a.hd;
            const float* ph = a.parts + (int64_t)h * NSPLIT * (a.hd + 4);      // 16-byte aligned slots: hd | max | denominator | pad
            float m = -INFINITY;
#pragma unroll
            for (int u = 0; u < NSPLIT; ++u) m = fmaxf(m, ph[u * (a.hd + 4) + a.hd]);
            float den = 0.f;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < NSPLIT; ++u) {
                const float* pu = ph + u * (a.hd + 4);
                const float wgt = __expf(pu[a.hd] - m);             // an empty split has max = -inf: weight 0
                const float4 ov = *reinterpret_cast<const float4*>(pu + cc);
                den += wgt * pu[a.hd + 1];
                o.x += wgt * ov.x; o.y += wgt * ov.y; o.z += wgt * ov.z; o.w += wgt * ov.w;
            }
            const float r = 1.f / den;
            xv[j] = make_float4(o.x * r, o.y * r, o.z * r, o.w * r);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) xv[j] = reinterpret_cast<const float4*>(a.x)[lane + 64 * j];
    }
    if (LN) {   // the arithmetic of norm_fwd_kernel<NV, false>, lane for lane
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += xv[j].x + xv[j].y + xv[j].z + xv[j].w;
        const float mean = dyn::wave_sum(s) / K;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float a0 = xv[j].x - mean, a1 = xv[j].y - mean, a2 = xv[j].z - mean, a3 = xv[j].w - mean;
            q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
        }
        const float rstd = rsqrtf(dyn::wave_sum(q) / K + a.eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 g = reinterpret_cast<const float4*>(a.gamma)[lane + 64 * j], b = reinterpret_cast<const float4*>(a.beta)[lane + 64 * j];
            xv[j].x = (xv[j].x - mean) * rstd * g.x + b.x;
            xv[j].y = (xv[j].y - mean) * rstd * g.y + b.y;
            xv[j].z = (xv[j].z - mean) * rstd * g.z + b.z;
            xv[j].w = (xv[j].w - mean) * rstd * g.w + b.w;
        }
    }
    const int n_base = (blockIdx.x * 4 + w) * a.rpw;
    for (int r = 0; r < a.rpw; r += 2) {          // two rows in flight per wave
        const int n0 = n_base + r;
        if (n0 >= a.N) break;
        const bool two = r + 1 < a.rpw && n0 + 1 < a.N;
        const float4* w0 = reinterpret_cast<const float4*>(a.W + (int64_t)n0 * K);
        const float4* w1 = two ? w0 + K / 4 : w0;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            s0 += dot4(w0[lane + 64 * j], xv[j]);
            s1 += dot4(w1[lane + 64 * j], xv[j]);
        }
        s0 = dyn::wave_sum(s0);
        s1 = dyn::wave_sum(s1);
        if (lane == 0) {
            float v = s0 + (a.bias ? a.bias[n0] : 0.f);
            if (SILU) v = v * dyn::sigmoidf_(v);
            a.y[n0] = v + (a.res ? a.res[n0] : 0.f);
            if (two) {
                v = s1 + (a.bias ? a.bias[n0 + 1] : 0.f);
                if (SILU) v = v * dyn::sigmoidf_(v);
                a.y[n0 + 1] = v + (a.res ? a.res[n0 + 1] : 0.f);
            }
        }
    }
}

struct AttnArgs {
    const float* q;    // [heads * hd]
    const float* k;    // row j of head h at k + j * ld + h * hd
    const float* v;
    float* parts;      // [heads][NSPLIT][hd + 4]: unnormalised weighted value sum | score maximum | sum of exponentials of this split's keys
    int n_keys, ld, hd;
    float scale;
};

// One query; workgroup (h, s) takes the s-th quarter of the keys of head h.  LDS: the split's scores, then 256 partial sums.
__global__ __launch_bounds__(256) void dec_attn_kernel(const AttnArgs a) {
    extern __shared__ float sc[];
    __shared__ float red[16];
    const int h = blockIdx.x, tid = threadIdx.x, g = tid >> 4, l = tid & 15;
    const int hd = a.hd;
    const int chunk = (a.n_keys + NSPLIT - 1) / NSPLIT;
    const int j_lo = blockIdx.y * chunk;
    const int n = min(chunk, a.n_keys - j_lo);       // keys of this split (<= 0: none)
    float* out = a.parts + ((int64_t)h * NSPLIT + blockIdx.y) * (hd + 4);
    if (n <= 0) {
        if (tid < hd) out[tid] = 0.f;
        if (tid == 0) { out[hd] = -INFINITY; out[hd + 1] = 0.f; }
        return;
    }
    const float* q = a.q + h * hd;
    const float* k = a.k + (int64_t)j_lo * a.ld + h * hd;
    const float* v = a.v + (int64_t)j_lo * a.ld + h * hd;
    float4 qv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = l * 4 + 64 * i;
        qv[i] = c < hd ? *reinterpret_cast<const float4*>(q + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // 16 lanes per key row, four rows in flight per 16-lane group (the loads of a group are independent: one latency per 64 keys)
    for (int j0 = g; j0 < n; j0 += 64) {
        float s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 16 * u;
            const float* kr = k + (int64_t)(j < n ? j : j0) * a.ld;
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = l * 4 + 64 * i;
                if (c < hd) t += dot4(*reinterpret_cast<const float4*>(kr + c), qv[i]);
            }
            s[u] = t;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float t = s[u];
            t += __shfl_xor(t, 8, 64);
            t += __shfl_xor(t, 4, 64);
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 1, 64);
            if (l == 0 && j0 + 16 * u < n) sc[j0 + 16 * u] = t * a.scale;
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = tid; j < n; j += 256) m = fmaxf(m, sc[j]);
    m = dyn::block_max(m, red);
    float sum = 0.f;
    for (int j = tid; j < n; j += 256) {
        const float e = __expf(sc[j] - m);
        sc[j] = e;
        sum += e;
    }
    sum = dyn::block_sum(sum, red);               // its barriers also publish the exponentials
    const int c = tid % hd, kg = tid / hd, ng = 256 / hd;
    float acc = 0.f;
    {   // eight value rows in flight per thread; the summation order (increasing j) does not depend on the unrolling
        int j = kg;
        for (; j + 7 * ng < n; j += 8 * ng) {
            float vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = v[(int64_t)(j + u * ng) * a.ld + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += sc[j + u * ng] * vv[u];
        }
        for (; j < n; j += ng) acc += sc[j] * v[(int64_t)j * a.ld + c];
    }
    float* part = sc + n;
    part[tid] = acc;
    __syncthreads();
    if (tid < hd) {
        float t = 0.f;
        for (int i = 0; i < ng; ++i) t += part[i * hd + tid];
        out[tid] = t;
    }
    if (tid == 0) { out[hd] = m; out[hd + 1] = sum; }
}

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// tokens_out[0] = argmax_c key(x[c]), first maximum wins; key = x (greedy) or the Gumbel-max key of dyn_gumbel_argmax_rows
template <bool SAMPLE>
__global__ __launch_bounds__(1024) void dec_pick_kernel(const float* __restrict__ x, int C, float inv_t, uint64_t seed, uint64_t stream,
                                                        int32_t* __restrict__ token_out) {
    __shared__ float rv[16];
    __shared__ int ri[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < C; c += 1024) {
        const float v = SAMPLE ? dyn::gumbel_key(x[c], inv_t, seed, stream, (uint64_t)c) : x[c];
        if (better(v, c, bv, bi)) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { rv[w] = bv; ri[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i)
            if (better(rv[i], ri[i], bv, bi)) { bv = rv[i]; bi = ri[i]; }
        token_out[0] = bi == 0x7fffffff ? 0 : bi;
    }
}

template <bool LN, bool SILU, bool EMBED, bool MERGE = false>
int launch_gemv(const GemvArgs& a, int K, hipStream_t st) {
    const dim3 grid((unsigned)dyn::cdiv(a.N, 4 * a.rpw)), blk(256);
    switch (K / 256) {
        case 1: hipLaunchKernelGGL((dec_gemv_kernel<1, LN, SILU, EMBED, MERGE>), grid, blk, 0, st, a); break;
        case 2: hipLaunchKernelGGL((dec_gemv_kernel<2, LN, SILU, EMBED, MERGE>), grid, blk, 0, st, a); break;
        case 3: hipLaunchKernelGGL((dec_gemv_kernel<3, LN, SILU, EMBED, MERGE>), grid, blk, 0, st, a); break;
        case 4: hipLaunchKernelGGL((dec_gemv_kernel<4, LN, SILU, EMBED, MERGE>), grid, blk, 0, st, a); break;
        case 5: hipLaunchKernelGGL((dec_gemv_kernel<5, LN, SILU, EMBED, MERGE>), grid, blk, 0, st, a); break;
        case 6: hipLaunchKernelGGL((dec_gemv_kernel<6, LN, SILU, EMBED, MERGE>), grid, blk, 0, st, a); break;
        case 7: hipLaunchKernelGGL((dec_gemv_kernel<7, LN, SILU, EMBED, MERGE>), grid, blk, 0, st, a); break;
        case 8: hipLaunchKernelGGL((dec_gemv_kernel<8, LN, SILU, EMBED, MERGE>), grid, blk, 0, st, a); break;
        default: return DYN_E_UNSUPPORTED;
    }
    return DYN_OK;
}

inline int rows_per_wave(int N) { return N >= 4096 ? 4 : (N >= 512 ? 2 : 1); }

GemvArgs gemv(const float* x, const float* W, const float* bias, const float* res, float* y, int N) {
    GemvArgs a{};
    a.x = x; a.W = W; a.bias = bias; a.res = res; a.y = y; a.N = N; a.rpw = rows_per_wave(N);
    return a;
}

constexpr int MAX_KEYS = 4 * 12288;   // (n_keys / NSPLIT + 256) floats of LDS <= 50 KB

}  // namespace

extern "C" int dyn_decoder_steps(const dyn_decoder_desc* d, int32_t t0, int32_t n_steps, int32_t sample, float inv_temperature,
                                 uint64_t seed, uint64_t step0, void* stream) {
    DYN_REQUIRE(d && d->embed && d->pos_table && d->norm_out_w && d->norm_out_b && d->head_w && d->head_b && d->layer_ptrs && d->tokens &&
                    d->logits && d->scratch, DYN_E_ARG, "dyn_decoder_steps: null pointer in the descriptor");
    const int dd = d->d_model, ff = d->d_ff, V = d->vocab, L = d->layers, H = d->heads;
    DYN_REQUIRE(dd > 0 && ff > 0 && V > 0 && L > 0 && H > 0 && d->n_enc > 0 && t0 >= 0 && n_steps >= 0, DYN_E_ARG, "dyn_decoder_steps: bad sizes");
    DYN_REQUIRE(dd % 256 == 0 && dd <= 2048 && ff % 256 == 0 && ff <= 2048, DYN_E_UNSUPPORTED,
                "dyn_decoder_steps: d_model %d / d_ff %d unsupported (multiples of 256 up to 2048)", dd, ff);
    DYN_REQUIRE(dd % H == 0, DYN_E_ARG, "dyn_decoder_steps: d_model %d not divisible by %d heads", dd, H);
    const int hd = dd / H;
    DYN_REQUIRE(hd >= 4 && hd <= 256 && (hd & (hd - 1)) == 0, DYN_E_UNSUPPORTED, "dyn_decoder_steps: head dim %d unsupported (power of two in 4 .. 256)", hd);
    DYN_REQUIRE((int64_t)t0 + n_steps <= d->max_positions, DYN_E_ARG, "dyn_decoder_steps: positions %d .. %d exceed max_positions %d", t0,
                t0 + n_steps - 1, d->max_positions);
    DYN_REQUIRE(d->n_enc <= MAX_KEYS && t0 + n_steps <= MAX_KEYS, DYN_E_UNSUPPORTED, "dyn_decoder_steps: more than %d keys per attention", MAX_KEYS);
    const int64_t need = (int64_t)2 * dd + ff + (int64_t)H * NSPLIT * (hd + 4);
    DYN_REQUIRE(d->scratch_floats >= need, DYN_E_WORKSPACE, "dyn_decoder_steps: scratch %lld < %lld floats", (long long)d->scratch_floats, (long long)need);
    DYN_REQUIRE(!sample || inv_temperature > 0.f, DYN_E_ARG, "dyn_decoder_steps: sampling needs a positive inverse temperature");
    for (int i = 0; i < L * DYN_DEC_PTRS_PER_LAYER; ++i)
        DYN_REQUIRE(d->layer_ptrs[i] != nullptr, DYN_E_ARG, "dyn_decoder_steps: layer pointer %d is null", i);
    hipStream_t st = (hipStream_t)stream;
    float* x = d->scratch;            // residual stream [dd]
    float* q2 = x + dd;               // cross-attention query [dd]
    float* act = q2 + dd;             // SiLU(w1 .) [ff]
    float* parts = act + ff;          // key splits of the attention [H][NSPLIT][hd + 4]
    auto merged = [&](const float* W, const float* bias) {      // x += W . attention + bias, the splits merged in the prologue
        GemvArgs m = gemv(nullptr, W, bias, x, x, dd);
        m.parts = parts; m.hd = hd;
        return m;
    };
    const float scale = 1.0f / sqrtf((float)hd);
    int rc = DYN_OK;
    for (int t = t0; t < t0 + n_steps && rc == DYN_OK; ++t) {
        for (int l = 0; l < L && rc == DYN_OK; ++l) {
            const void* const* P = d->layer_ptrs + (size_t)l * DYN_DEC_PTRS_PER_LAYER;
            auto F = [&](int i) { return (const float*)P[i]; };
            float* cache = (float*)P[16];
            const float* ckv = F(17);
            float* row = cache + (int64_t)t * 3 * dd;
            // self-attention: q | k | v of this position into cache row t, then the query against rows 0 .. t
            GemvArgs a = gemv(x, F(2), F(3), nullptr, row, 3 * dd);
            a.gamma = F(0); a.beta = F(1); a.eps = d->eps;
            if (l == 0) {
                a.tok = d->tokens + t; a.table = d->embed; a.pos = d->pos_table + (int64_t)t * dd; a.x_out = x; a.vocab = V;
                rc = launch_gemv<true, false, true>(a, dd, st);
            } else {
                rc = launch_gemv<true, false, false>(a, dd, st);
            }
            if (rc != DYN_OK) break;
            AttnArgs s{row, cache + dd, cache + 2 * dd, parts, t + 1, 3 * dd, hd, scale};
            hipLaunchKernelGGL(dec_attn_kernel, dim3(H, NSPLIT), dim3(256), (size_t)(dyn::cdiv(t + 1, NSPLIT) + 256) * sizeof(float), st, s);
            rc = launch_gemv<false, false, false, true>(merged(F(4), F(5)), dd, st);
            if (rc != DYN_OK) break;
            // cross-attention over the projected encoder rows
            a = gemv(x, F(8), F(9), nullptr, q2, dd);
            a.gamma = F(6); a.beta = F(7); a.eps = d->eps;
            rc = launch_gemv<true, false, false>(a, dd, st);
            if (rc != DYN_OK) break;
            AttnArgs c{q2, ckv, ckv + dd, parts, d->n_enc, 2 * dd, hd, scale};
            hipLaunchKernelGGL(dec_attn_kernel, dim3(H, NSPLIT), dim3(256), (size_t)(dyn::cdiv(d->n_enc, NSPLIT) + 256) * sizeof(float), st, c);
            rc = launch_gemv<false, false, false, true>(merged(F(10), F(11)), dd, st);
            if (rc != DYN_OK) break;
            // feed-forward
            a = gemv(x, F(14), nullptr, nullptr, act, ff);
            a.gamma = F(12); a.beta = F(13); a.eps = d->eps;
            rc = launch_gemv<true, true, false>(a, dd, st);
            if (rc != DYN_OK) break;
            rc = launch_gemv<false, false, false>(gemv(act, F(15), nullptr, x, x, dd), ff, st);
        }
        if (rc != DYN_OK) break;
        GemvArgs a = gemv(x, d->head_w, d->head_b, nullptr, d->logits, V);
        a.gamma = d->norm_out_w; a.beta = d->norm_out_b; a.eps = d->eps;
        rc = launch_gemv<true, false, false>(a, dd, st);
        if (rc != DYN_OK) break;
        if (sample)
            hipLaunchKernelGGL((dec_pick_kernel<true>), dim3(1), dim3(1024), 0, st, d->logits, V, inv_temperature, seed, step0 + (uint64_t)t, d->tokens + t + 1);
        else
            hipLaunchKernelGGL((dec_pick_kernel<false>), dim3(1), dim3(1024), 0, st, d->logits, V, 1.f, 0ull, 0ull, d->tokens + t + 1);
    }
    DYN_REQUIRE(rc == DYN_OK, rc, "dyn_decoder_steps: unsupported row length");
    return dyn::check_launch("dyn_decoder_steps");
}
