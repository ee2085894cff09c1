// Row softmax / log-softmax forward + backward (rows x L, any L <= 16384): one 256-thread workgroup per row,
// the row is read ONCE into registers (ITEMS values per thread, coalesced stride-256 accesses), max and sum are
// wavefront + LDS reductions, the result is written once: 1 read + 1 write of HBM per element.
// Reference call sites: attention softmax and the CTC head's log_softmax inside model(audio_signal=...)
// (reference lcasr/lib.py:550), `F.log_softmax` on wav2vec2 logits (reference wav2vec2/lib.py:169,417),
// the self-conditioning softmax (yaml `self_conditioning: true`), and their backward passes (lib.py:579).
#include "common.h"

namespace {

constexpr int TPB = 256;

template <int ITEMS, bool LOG>
__global__ __launch_bounds__(TPB) void softmax_fwd_kernel(const float* x, float* y, int64_t rows, int L, int64_t ldx,
                                                           int64_t ldy, const int32_t* __restrict__ valid) {  // y may alias x (in-place attention softmax)
    __shared__ float red[8];
    // `valid` (device scalar, may be null): columns >= *valid are masked keys — outside the max and the sum, written as 0 (-inf for LOG).
    // The values of the first *valid columns are bit for bit those of a row of length *valid (same per-thread items, same reductions).
    int Lv = L;
    if (valid) { const int v = *valid; Lv = v < 1 ? 1 : (v < L ? v : L); }
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* xr = x + row * ldx;
        float v[ITEMS];
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int c = threadIdx.x + j * TPB;
            v[j] = c < Lv ? xr[c] : -INFINITY;
            m = fmaxf(m, v[j]);
        }
        m = dyn::block_max(m, red);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int c = threadIdx.x + j * TPB;
            const float e = c < Lv ? __expf(v[j] - m) : 0.f;
            s += e;
            if (!LOG) v[j] = e;
        }
        s = dyn::block_sum(s, red);
        float* yr = y + row * ldy;
        if (LOG) {
            const float lse = m + __logf(s);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int c = threadIdx.x + j * TPB;
                if (c < L) yr[c] = v[j] - lse;
            }
        } else {
            const float inv = 1.f / s;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int c = threadIdx.x + j * TPB;
                if (c < L) yr[c] = v[j] * inv;
            }
        }
    }
}

// softmax:      dx = y * (dy - sum(dy * y)) * scale
// log_softmax:  dx = dy - exp(y) * sum(dy)
template <int ITEMS, bool LOG>
__global__ __launch_bounds__(TPB) void softmax_bwd_kernel(const float* __restrict__ y, const float* dy, float* dx,
                                                           int64_t rows, int L, int64_t ld, float scale) {
    __shared__ float red[8];
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* yr = y + row * ld;
        const float* gr = dy + row * ld;
        float yv[ITEMS], gv[ITEMS];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int c = threadIdx.x + j * TPB;
            yv[j] = c < L ? yr[c] : 0.f;
            gv[j] = c < L ? gr[c] : 0.f;
            s += LOG ? gv[j] : gv[j] * yv[j];
        }
        s = dyn::block_sum(s, red);
        float* xr = dx + row * ld;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int c = threadIdx.x + j * TPB;
            if (c < L) xr[c] = LOG ? (gv[j] - __expf(yv[j]) * s) : (yv[j] * (gv[j] - s) * scale);
        }
    }
}

// Gradient of the MEAN categorical entropy w.r.t. log-probabilities y (rows already normalised):
//   H_row = -sum_c p_c y_c,  p = exp(y);   dH_row/dy_c = -p_c (y_c + H_row)   (the simplex projection included);
//   grad = scale * dH_row/dy  with scale = 1 / rows for `entropy.mean()`  (reference lcasr/lib.py:94-96).
template <int ITEMS>
__global__ __launch_bounds__(TPB) void entropy_grad_kernel(const float* __restrict__ y, float* __restrict__ g,
                                                            float* __restrict__ ent, int64_t rows, int L, int64_t ld,
                                                            float scale) {
    __shared__ float red[8];
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* yr = y + row * ld;
        float yv[ITEMS], pv[ITEMS];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int c = threadIdx.x + j * TPB;
            yv[j] = c < L ? yr[c] : 0.f;
            pv[j] = c < L ? expf(yv[j]) : 0.f;
            s -= pv[j] * yv[j];
        }
        const float H = dyn::block_sum(s, red);
        float* gr = g + row * ld;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int c = threadIdx.x + j * TPB;
            if (c < L) gr[c] = -pv[j] * (yv[j] + H) * scale;
        }
        if (ent && threadIdx.x == 0) ent[row] = H;
    }
}

template <bool LOG>
int launch_fwd(const float* x, float* y, int64_t rows, int64_t L, int64_t ldx, int64_t ldy, hipStream_t st, const int32_t* valid = nullptr) {
    int64_t g = rows < 65535 * 4 ? rows : 65535 * 4;
    dim3 grid((unsigned)g), blk(TPB);
    const int items = (int)dyn::cdiv(L, TPB);
#define GO(I) hipLaunchKernelGGL((softmax_fwd_kernel<I, LOG>), grid, blk, 0, st, x, y, rows, (int)L, ldx, ldy, valid)
    if (items <= 1) GO(1);
    else if (items <= 2) GO(2);
    else if (items <= 4) GO(4);
    else if (items <= 8) GO(8);
    else if (items <= 16) GO(16);
    else if (items <= 32) GO(32);
    else if (items <= 64) GO(64);
    else { dyn::set_error("softmax: row length %lld > 16384 unsupported", (long long)L); return DYN_E_UNSUPPORTED; }
#undef GO
    return dyn::check_launch("dyn_softmax_fwd");
}

template <bool LOG>
int launch_bwd(const float* y, const float* dy, float* dx, int64_t rows, int64_t L, int64_t ld, float scale, hipStream_t st) {
    int64_t g = rows < 65535 * 4 ? rows : 65535 * 4;
    dim3 grid((unsigned)g), blk(TPB);
    const int items = (int)dyn::cdiv(L, TPB);
#define GO(I) hipLaunchKernelGGL((softmax_bwd_kernel<I, LOG>), grid, blk, 0, st, y, dy, dx, rows, (int)L, ld, scale)
    if (items <= 1) GO(1);
    else if (items <= 2) GO(2);
    else if (items <= 4) GO(4);
    else if (items <= 8) GO(8);
    else if (items <= 16) GO(16);
    else if (items <= 32) GO(32);
    else { dyn::set_error("softmax_bwd: row length %lld > 8192 unsupported", (long long)L); return DYN_E_UNSUPPORTED; }
#undef GO
    return dyn::check_launch("dyn_softmax_bwd");
}

}  // namespace

extern "C" int dyn_softmax_fwd(const float* x, float* y, int64_t rows, int64_t L, int64_t ldx, int64_t ldy, void* stream) {
    DYN_REQUIRE(x && y && rows >= 0 && L > 0 && ldx >= L && ldy >= L, DYN_E_ARG, "dyn_softmax_fwd: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_fwd<false>(x, y, rows, L, ldx, ldy, (hipStream_t)stream);
}
// Key-length masked softmax: `valid_cols` is a DEVICE int32 scalar read when the kernel runs (a captured launch serves every utterance
// length of a bucket); columns >= *valid_cols get probability 0.
extern "C" int dyn_softmax_fwd_len(const float* x, float* y, int64_t rows, int64_t L, int64_t ldx, int64_t ldy, const int32_t* valid_cols,
                                   void* stream) {
    DYN_REQUIRE(x && y && valid_cols && rows >= 0 && L > 0 && ldx >= L && ldy >= L, DYN_E_ARG, "dyn_softmax_fwd_len: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_fwd<false>(x, y, rows, L, ldx, ldy, (hipStream_t)stream, valid_cols);
}
extern "C" int dyn_softmax_bwd(const float* y, const float* dy, float* dx, int64_t rows, int64_t L, int64_t ld, float scale,
                               void* stream) {
    DYN_REQUIRE(y && dy && dx && rows >= 0 && L > 0 && ld >= L, DYN_E_ARG, "dyn_softmax_bwd: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_bwd<false>(y, dy, dx, rows, L, ld, scale, (hipStream_t)stream);
}
extern "C" int dyn_log_softmax_fwd(const float* x, float* y, int64_t rows, int64_t L, int64_t ldx, int64_t ldy, void* stream) {
    DYN_REQUIRE(x && y && rows >= 0 && L > 0 && ldx >= L && ldy >= L, DYN_E_ARG, "dyn_log_softmax_fwd: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_fwd<true>(x, y, rows, L, ldx, ldy, (hipStream_t)stream);
}
extern "C" int dyn_entropy_grad(const float* log_probs, float* grad, float* entropy_per_row, int64_t rows, int64_t L, int64_t ld,
                                float scale, void* stream) {
    DYN_REQUIRE(log_probs && grad && rows >= 0 && L > 0 && ld >= L, DYN_E_ARG, "dyn_entropy_grad: bad arguments");
    if (rows == 0) return DYN_OK;
    int64_t gq = rows < 65535 * 4 ? rows : 65535 * 4;
    dim3 grid((unsigned)gq), blk(TPB);
    hipStream_t st = (hipStream_t)stream;
    const int items = (int)dyn::cdiv(L, TPB);
#define GO(I) hipLaunchKernelGGL((entropy_grad_kernel<I>), grid, blk, 0, st, log_probs, grad, entropy_per_row, rows, (int)L, ld, scale)
    if (items <= 1) GO(1);
    else if (items <= 2) GO(2);
    else if (items <= 4) GO(4);
    else if (items <= 8) GO(8);
    else if (items <= 16) GO(16);
    else if (items <= 32) GO(32);
    else { dyn::set_error("entropy_grad: row length %lld > 8192 unsupported", (long long)L); return DYN_E_UNSUPPORTED; }
#undef GO
    return dyn::check_launch("dyn_entropy_grad");
}

extern "C" int dyn_log_softmax_bwd(const float* y, const float* dy, float* dx, int64_t rows, int64_t L, int64_t ld,
                                   void* stream) {
    DYN_REQUIRE(y && dy && dx && rows >= 0 && L > 0 && ld >= L, DYN_E_ARG, "dyn_log_softmax_bwd: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_bwd<true>(y, dy, dx, rows, L, ld, 1.f, (hipStream_t)stream);
}
