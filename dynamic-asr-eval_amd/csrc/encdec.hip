// Small kernels of the encoder-decoder `teacher_ce` adaptation path (reference lcasr/lib.py:1228-1322 `calc_loss_enc_dec`,
// :1475-1732 `enc_dec_dynamic_eval`): token-embedding gather and its deterministic gradient, the causal mask of the decoder's
// self-attention scores, and cross-entropy (sum over rows, ignore_index) with its gradient w.r.t. the logits.  Everything dense in
// the decoder (projections, attention products, FFN) goes through dyn_gemm_f32; norms / softmax / SiLU reuse the encoder's kernels.
#include "common.h"

namespace {

// out[s, :] = table[ids[s], :]  (+ pos[s, :] if given)
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int32_t* __restrict__ ids, const float* __restrict__ table,
                                                            const float* __restrict__ pos, float* __restrict__ out, int64_t S, int d,
                                                            int64_t pos_period, int64_t vocab) {
    const int64_t s = blockIdx.x;
    // ids are validated on the host (enc_dec.py: _check_ids) before they are uploaded; the clamp only guarantees that an id
    // from a foreign caller can never read outside the table
    int64_t id = ids[s];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* row = table + id * d;
    const float* pr = pos ? pos + (s % pos_period) * d : nullptr;
    for (int c = threadIdx.x; c < d; c += blockDim.x) out[s * d + c] = row[c] + (pr ? pr[c] : 0.f);
}

// dtable[v, :] = beta * dtable[v, :] + sum over positions s with ids[s] == v of dy[s, :], positions visited in increasing s:
// one workgroup per vocabulary row, no atomics (deterministic).  S is a few hundred tokens at most.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dy,
                                                            float* __restrict__ dtable, int64_t S, int d, float beta) {
    const int64_t v = blockIdx.x;
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
        float acc = 0.f;
        for (int64_t s = 0; s < S; ++s)
            if (ids[s] == v) acc += dy[s * d + c];
        dtable[v * d + c] = (beta != 0.f ? beta * dtable[v * d + c] : 0.f) + acc;
    }
}

// scores [nb, S, S]: entries with column > row become -inf (causal self-attention)
__global__ __launch_bounds__(256) void causal_mask_kernel(float* __restrict__ scores, int64_t nb, int64_t S) {
    const int64_t total = nb * S * S;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col = idx % S, row = (idx / S) % S;
        if (col > row) scores[idx] = -INFINITY;
    }
}

// Per row r: loss_r = -logp[r, target_r] (0 if target_r == ignore_index); grad[r, :] = scale * (exp(logp[r, :]) - onehot(target_r)),
// zero for ignored rows.  logp are log-softmax outputs.  row_loss [rows] is summed by the caller's deterministic reduction.
__global__ __launch_bounds__(256) void nll_grad_kernel(const float* __restrict__ logp, const int32_t* __restrict__ targets,
                                                       float* __restrict__ row_loss, float* __restrict__ grad, int64_t rows, int C,
                                                       int ignore_index, float scale) {
    const int64_t r = blockIdx.x;
    const int t = targets[r];
    const bool live = t != ignore_index && t >= 0 && t < C;     // an out-of-range target contributes nothing instead of reading out of bounds
    if (threadIdx.x == 0) row_loss[r] = live ? -logp[r * C + t] : 0.f;
    if (grad == nullptr) return;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float g = 0.f;
        if (live) g = scale * (__expf(logp[r * C + c]) - (c == t ? 1.f : 0.f));
        grad[r * C + c] = g;
    }
}

// ---- counter-based randomness: a draw is a pure function of (seed, stream, index); see include/dyneval.h (dyn::mix64 in common.h)
using dyn::mix64;

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p, float scale,
                                                      uint64_t seed, uint64_t stream) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float u = (float)(mix64(seed, stream, (uint64_t)i) >> 40) * 5.9604644775390625e-8f;   // 24 bits * 2^-24: exact in fp32
        y[i] = u >= p ? x[i] * scale : 0.f;
    }
}

// ids[row] = argmax_c (x[row, c] * inv_t + gumbel(seed, step0 + row, c)), first maximum wins
__global__ __launch_bounds__(256) void gumbel_argmax_rows_kernel(const float* __restrict__ x, int32_t* __restrict__ ids, int64_t rows, int C,
                                                                 int64_t ld, float inv_t, uint64_t seed, uint64_t step0) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < rows; row += (int64_t)gridDim.x * 4) {
        const float* xr = x + row * ld;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float v = dyn::gumbel_key(xr[c], inv_t, seed, step0 + (uint64_t)row, (uint64_t)c);
            if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) ids[row] = bi == 0x7fffffff ? 0 : bi;
    }
}

__global__ void sum_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += x[i];      // fixed assignment of elements to threads: deterministic
    s = dyn::block_sum(s, red);
    if (threadIdx.x == 0) *out = s;
}

}  // namespace

extern "C" int dyn_embedding_fwd(const int32_t* ids, const float* table, const float* pos, float* out, int64_t S, int64_t d,
                                 int64_t vocab, int64_t pos_period, void* stream) {
    DYN_REQUIRE(ids && table && out && S >= 0 && d > 0 && vocab > 0 && (!pos || pos_period > 0), DYN_E_ARG, "dyn_embedding_fwd: bad arguments");
    if (S == 0) return DYN_OK;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3((unsigned)S), dim3(256), 0, (hipStream_t)stream, ids, table, pos, out, S, (int)d,
                       pos ? pos_period : 1, vocab);
    return dyn::check_launch("dyn_embedding_fwd");
}

extern "C" int dyn_embedding_bwd(const int32_t* ids, const float* dy, float* dtable, int64_t S, int64_t d, int64_t vocab, float beta,
                                 void* stream) {
    DYN_REQUIRE(ids && dy && dtable && S >= 0 && d > 0 && vocab > 0, DYN_E_ARG, "dyn_embedding_bwd: bad arguments");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)vocab), dim3(256), 0, (hipStream_t)stream, ids, dy, dtable, S, (int)d, beta);
    return dyn::check_launch("dyn_embedding_bwd");
}

extern "C" int dyn_causal_mask(float* scores, int64_t nb, int64_t S, void* stream) {
    DYN_REQUIRE(scores && nb >= 0 && S >= 0, DYN_E_ARG, "dyn_causal_mask: bad arguments");
    if (nb * S * S == 0) return DYN_OK;
    int64_t g = dyn::cdiv(nb * S * S, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(causal_mask_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, scores, nb, S);
    return dyn::check_launch("dyn_causal_mask");
}

extern "C" int dyn_nll_loss(const float* log_probs, const int32_t* targets, float* loss, float* row_loss, float* grad, int64_t rows,
                            int64_t C, int32_t ignore_index, float grad_scale, void* stream) {
    DYN_REQUIRE(log_probs && targets && loss && row_loss && rows >= 0 && C > 0, DYN_E_ARG, "dyn_nll_loss: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (rows > 0)
        hipLaunchKernelGGL(nll_grad_kernel, dim3((unsigned)rows), dim3(256), 0, st, log_probs, targets, row_loss, grad, rows, (int)C,
                           (int)ignore_index, grad_scale);
    hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(256), 0, st, row_loss, loss, rows);
    return dyn::check_launch("dyn_nll_loss");
}

extern "C" int dyn_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream) {
    DYN_REQUIRE(x && y && n >= 0 && p >= 0.f && p < 1.f, DYN_E_ARG, "dyn_dropout: need 0 <= p < 1");
    if (n == 0) return DYN_OK;
    int64_t g = dyn::cdiv(n, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, y, n, p, 1.0f / (1.0f - p), seed, stream_id);
    return dyn::check_launch("dyn_dropout");
}

extern "C" int dyn_gumbel_argmax_rows(const float* x, int64_t rows, int64_t C, int64_t ld, float inv_temperature, uint64_t seed,
                                      uint64_t step0, int32_t* ids, void* stream) {
    DYN_REQUIRE(x && ids && rows >= 0 && C > 0 && ld >= C && inv_temperature > 0.f, DYN_E_ARG, "dyn_gumbel_argmax_rows: bad arguments");
    if (rows == 0) return DYN_OK;
    int64_t g = dyn::cdiv(rows, 4);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(gumbel_argmax_rows_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, ids, rows, (int)C, ld,
                       inv_temperature, seed, step0);
    return dyn::check_launch("dyn_gumbel_argmax_rows");
}
