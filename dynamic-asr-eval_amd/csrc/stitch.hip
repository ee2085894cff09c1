// Window stitching on device: exp -> overlap-add -> count -> divide -> log, and rotary position embedding.
//   stitch: reference lcasr/lib.py:583-589,604-609 (exp of each window's log-posteriors) and :615-629
//           (position bookkeeping, sum / count, log).  The reference keeps two [T/4 + seq_len, V+1] fp32 buffers
//           on the HOST and copies every window's posteriors over PCIe; here the accumulators live in HBM and
//           the only thing that ever leaves the device is the final [T_ds, V+1] result (or just its argmax ids).
//   rotary: `use_rotary: true, rotary_base_freq: 1500000` (earnings_finetune/lcasr160rb1.yaml:22,28), applied to
//           q and k in place inside the packed QKV activation.
#include "common.h"

namespace {

constexpr int TPB = 256;

// acc[pos + r, c] += exp(lp[r, c]);  cnt[pos + r] += 1
__global__ __launch_bounds__(TPB) void stitch_accumulate_kernel(const float* __restrict__ lp, int64_t ld, float* __restrict__ acc,
                                                                float* __restrict__ cnt, int64_t pos, int64_t rows, int C) {
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* src = lp + r * ld;
        float* dst = acc + (pos + r) * C;
        for (int c = threadIdx.x; c < C; c += TPB) dst[c] += expf(src[c]);
        if (threadIdx.x == 0) cnt[pos + r] += 1.f;
    }
}

// out[r, c] = log(acc[r, c] / cnt[r])   (rows with cnt == 0 are never part of the output: the host passes the
// covered prefix only, see lib.py:624-627 of the reference)
__global__ __launch_bounds__(TPB) void stitch_finalize_kernel(const float* __restrict__ acc, const float* __restrict__ cnt,
                                                              float* __restrict__ out, int64_t rows, int C) {
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float n = cnt[r];
        const float* src = acc + r * C;
        float* dst = out + r * C;
        for (int c = threadIdx.x; c < C; c += TPB) dst[c] = logf(src[c] / n);
    }
}

// out[r, c] = log(acc[idx[r], c] / cnt[idx[r]]): the covered rows of an accumulator whose coverage has gaps
// (audio-disjoint leave-one-out stitching, reference lcasr/run_within_recording_loo_eval.py:160-181)
__global__ __launch_bounds__(TPB) void stitch_finalize_rows_kernel(const float* __restrict__ acc, const float* __restrict__ cnt,
                                                                   const int64_t* __restrict__ idx, float* __restrict__ out,
                                                                   int64_t rows, int C) {
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const int64_t g = idx[r];
        const float n = cnt[g];
        const float* src = acc + g * C;
        float* dst = out + r * C;
        for (int c = threadIdx.x; c < C; c += TPB) dst[c] = logf(src[c] / n);
    }
}

// x: rows of `row_stride` floats, the first n_heads * D of each row are [head][D] blocks to rotate (q and k).
// (x1, x2) = (x[i], x[i + D/2]);  fwd: (x1 c - x2 s, x2 c + x1 s);  bwd (transpose): (g1 c + g2 s, g2 c - g1 s).
__global__ __launch_bounds__(TPB) void rotary_kernel(float* __restrict__ x, const float* __restrict__ cos_t,
                                                     const float* __restrict__ sin_t, int64_t B, int64_t T, int n_heads, int D,
                                                     int64_t row_stride, int inverse) {
    const int half = D >> 1;
    const int64_t per_row = (int64_t)n_heads * half;
    const int64_t total = B * T * per_row;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int64_t row = i / per_row;
        const int rem = (int)(i % per_row);
        const int h = rem / half, k = rem % half;
        const int64_t t = row % T;
        const float c = cos_t[t * half + k];
        const float s = inverse ? -sin_t[t * half + k] : sin_t[t * half + k];
        float* p = x + row * row_stride + (int64_t)h * D + k;
        const float x1 = p[0], x2 = p[half];
        p[0] = x1 * c - x2 * s;
        p[half] = x2 * c + x1 * s;
    }
}

}  // namespace

extern "C" int dyn_stitch_accumulate(const float* log_probs, int64_t ld, float* acc, float* count, int64_t pos, int64_t rows,
                                     int64_t C, int64_t acc_rows, void* stream) {
    DYN_REQUIRE(log_probs && acc && count && rows >= 0 && C > 0 && ld >= C && pos >= 0 && pos + rows <= acc_rows, DYN_E_ARG,
                "dyn_stitch_accumulate: window [%lld, %lld) outside the %lld-row accumulator", (long long)pos,
                (long long)(pos + rows), (long long)acc_rows);
    if (rows == 0) return DYN_OK;
    hipLaunchKernelGGL(stitch_accumulate_kernel, dim3((unsigned)(rows < 4096 ? rows : 4096)), dim3(TPB), 0, (hipStream_t)stream,
                       log_probs, ld, acc, count, pos, rows, (int)C);
    return dyn::check_launch("dyn_stitch_accumulate");
}

extern "C" int dyn_stitch_finalize(const float* acc, const float* count, float* out, int64_t rows, int64_t C, void* stream) {
    DYN_REQUIRE(acc && count && out && rows >= 0 && C > 0, DYN_E_ARG, "dyn_stitch_finalize: bad arguments");
    if (rows == 0) return DYN_OK;
    hipLaunchKernelGGL(stitch_finalize_kernel, dim3((unsigned)(rows < 8192 ? rows : 8192)), dim3(TPB), 0, (hipStream_t)stream, acc,
                       count, out, rows, (int)C);
    return dyn::check_launch("dyn_stitch_finalize");
}

extern "C" int dyn_stitch_finalize_rows(const float* acc, const float* count, const int64_t* row_index, float* out, int64_t rows,
                                        int64_t C, void* stream) {
    DYN_REQUIRE(acc && count && out && rows >= 0 && C > 0 && (rows == 0 || row_index), DYN_E_ARG, "dyn_stitch_finalize_rows: bad arguments");
    if (rows == 0) return DYN_OK;
    hipLaunchKernelGGL(stitch_finalize_rows_kernel, dim3((unsigned)(rows < 8192 ? rows : 8192)), dim3(TPB), 0, (hipStream_t)stream,
                       acc, count, row_index, out, rows, (int)C);
    return dyn::check_launch("dyn_stitch_finalize_rows");
}

extern "C" int dyn_rotary(float* x, const float* cos_table, const float* sin_table, int64_t B, int64_t T, int64_t n_heads,
                          int64_t D, int64_t row_stride, int32_t inverse, void* stream) {
    DYN_REQUIRE(x && cos_table && sin_table && B >= 0 && T >= 0 && n_heads > 0 && D > 0 && D % 2 == 0 && row_stride >= n_heads * D,
                DYN_E_ARG, "dyn_rotary: bad arguments");
    const int64_t total = B * T * n_heads * (D / 2);
    if (total == 0) return DYN_OK;
    int64_t g = dyn::cdiv(total, TPB);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(rotary_kernel, dim3((unsigned)g), dim3(TPB), 0, (hipStream_t)stream, x, cos_table, sin_table, B, T,
                       (int)n_heads, (int)D, row_stride, (int)inverse);
    return dyn::check_launch("dyn_rotary");
}
