// Optional window augmentations of the reference's Loop A (reference lcasr/lib.py:542-544), on the device-resident
// [F, T] window; every random draw (permutations, rectangles, the noise tensor itself) stays on the host so the RNG
// stream is the caller's, exactly as in the reference where these run on CPU tensors before `.to(device)`:
//   frame_shuffle   lib.py:81-84    y[f, t] = x[f, perm_t[t]]  /  y[f, t] = x[perm_f[f], t]
//   add_random_noise lib.py:379-382 needs spec.std(): dyn_moments gives (sum, sum of squares) in a fixed order
//   cutout          lib.py:384-417  rectangles filled with their own mean / the window mean / zero
#include "common.h"

namespace {
constexpr int TPB = 256;

__global__ __launch_bounds__(TPB) void gather_cols_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                          float* __restrict__ y, int F, int64_t T) {
    const int f = blockIdx.y;
    for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < T; t += (int64_t)gridDim.x * TPB)
        y[(int64_t)f * T + t] = x[(int64_t)f * T + idx[t]];
}

__global__ __launch_bounds__(TPB) void gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                          float* __restrict__ y, int F, int64_t T) {
    const int f = blockIdx.y;
    const int64_t src = (int64_t)idx[f] * T;
    for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < T; t += (int64_t)gridDim.x * TPB) y[(int64_t)f * T + t] = x[src + t];
}

// partial[block] = (sum, sumsq) of a grid-stride slice; the host-visible result is reduced by moments_final_kernel.
__global__ __launch_bounds__(TPB) void moments_partial_kernel(const float* __restrict__ x, double* __restrict__ partial, int64_t n) {
    __shared__ float red[8];
    float s = 0.f, q = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const float v = x[i];
        s += v;
        q += v * v;
    }
    s = dyn::block_sum(s, red);
    q = dyn::block_sum(q, red);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = (double)s; partial[2 * blockIdx.x + 1] = (double)q; }
}

// one wave: lane l adds the partials l, l + 64, ... (independent loads), then a fixed butterfly over the lanes — deterministic
__global__ void moments_final_kernel(const double* __restrict__ partial, int nb, int64_t n, float* __restrict__ out3) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nb; i += 64) { s += partial[2 * i]; q += partial[2 * i + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if (threadIdx.x != 0) return;
    const double mean = s / (double)n;
    const double var = n > 1 ? (q - s * mean) / (double)(n - 1) : 0.0;  // unbiased, as torch.std()
    out3[0] = (float)s;
    out3[1] = (float)mean;
    out3[2] = (float)sqrt(var > 0.0 ? var : 0.0);
}

// rect = (y0, y1, x0, x1); mean over the rectangle (one workgroup per rectangle)
__global__ __launch_bounds__(TPB) void rect_mean_kernel(const float* __restrict__ x, int64_t T, const int32_t* __restrict__ rects,
                                                        float* __restrict__ means) {
    __shared__ float red[8];
    const int32_t* r = rects + 4 * blockIdx.x;
    const int h = r[1] - r[0], w = r[3] - r[2];
    float s = 0.f;
    for (int e = threadIdx.x; e < h * w; e += TPB) s += x[(int64_t)(r[0] + e / w) * T + r[2] + e % w];
    s = dyn::block_sum(s, red);
    if (threadIdx.x == 0) means[blockIdx.x] = (h * w) > 0 ? s / (float)(h * w) : 0.f;
}

// rectangles are filled in order (later rectangles overwrite earlier ones, as the sequential host loop does)
__global__ __launch_bounds__(TPB) void rect_fill_kernel(float* __restrict__ x, int64_t T, const int32_t* __restrict__ rects,
                                                        const float* __restrict__ means, int n_rects, float value, int use_means) {
    for (int k = 0; k < n_rects; ++k) {
        const int32_t* r = rects + 4 * k;
        const int h = r[1] - r[0], w = r[3] - r[2];
        const float v = use_means ? means[k] : value;
        for (int e = blockIdx.x * TPB + threadIdx.x; e < h * w; e += gridDim.x * TPB) x[(int64_t)(r[0] + e / w) * T + r[2] + e % w] = v;
        __syncthreads();
    }
}
}  // namespace

extern "C" int dyn_gather_frames(const float* x, const int32_t* index, float* y, int64_t F, int64_t T, int32_t along_time,
                                 void* stream) {
    DYN_REQUIRE(x && index && y && x != y && F > 0 && T > 0 && F < 65536, DYN_E_ARG, "dyn_gather_frames: bad arguments");
    int64_t gx = dyn::cdiv(T, TPB * 4);
    if (gx > 1024) gx = 1024;
    if (along_time) hipLaunchKernelGGL(gather_cols_kernel, dim3((unsigned)gx, (unsigned)F), dim3(TPB), 0, (hipStream_t)stream, x, index, y, (int)F, T);
    else hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)gx, (unsigned)F), dim3(TPB), 0, (hipStream_t)stream, x, index, y, (int)F, T);
    return dyn::check_launch("dyn_gather_frames");
}

extern "C" int64_t dyn_moments_workspace_bytes(void) { return 2 * 1024 * (int64_t)sizeof(double); }

// out3 = (sum, mean, unbiased std) of x[0..n)
extern "C" int dyn_moments(const float* x, int64_t n, float* out3, void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(x && out3 && n > 0, DYN_E_ARG, "dyn_moments: bad arguments");
    DYN_REQUIRE(workspace && workspace_bytes >= dyn_moments_workspace_bytes(), DYN_E_WORKSPACE, "dyn_moments: workspace too small");
    int nb = (int)dyn::cdiv(n, TPB * 8);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(moments_partial_kernel, dim3(nb), dim3(TPB), 0, st, x, (double*)workspace, n);
    hipLaunchKernelGGL(moments_final_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, nb, n, out3);
    return dyn::check_launch("dyn_moments");
}

// mode: 0 = 'zero', 1 = 'mean' (each rectangle's own mean, all means taken before any fill), 2 = constant `value`
extern "C" int dyn_cutout(float* x, int64_t F, int64_t T, const int32_t* rects, int64_t n_rects, int32_t mode, float value,
                          float* means_scratch, void* stream) {
    DYN_REQUIRE(x && F > 0 && T > 0 && n_rects >= 0 && (n_rects == 0 || rects) && (mode != 1 || means_scratch), DYN_E_ARG,
                "dyn_cutout: bad arguments");
    if (n_rects == 0) return DYN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (mode == 1) hipLaunchKernelGGL(rect_mean_kernel, dim3((unsigned)n_rects), dim3(TPB), 0, st, x, T, rects, means_scratch);
    hipLaunchKernelGGL(rect_fill_kernel, dim3(1), dim3(TPB), 0, st, x, T, rects, means_scratch, (int)n_rects, mode == 0 ? 0.f : value,
                       mode == 1 ? 1 : 0);
    return dyn::check_launch("dyn_cutout");
}


// ---- device-side delay: one wave waits on the constant-rate (100 MHz) realtime counter.  Used by lib.dynamic_eval_many to start the
// recording chains out of phase (identical chains started together stay phase-locked: all of them run their HBM-bound / latency-bound
// stretches at the same time and the matrix cores idle).
namespace {
__global__ void sleep_kernel(long long ticks) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}
}  // namespace

extern "C" int dyn_sleep_us(int64_t microseconds, void* stream) {
    DYN_REQUIRE(microseconds >= 0 && microseconds <= 2000000, DYN_E_ARG, "dyn_sleep_us: 0 <= microseconds <= 2e6");
    if (microseconds == 0) return DYN_OK;
    hipLaunchKernelGGL(sleep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100);
    return dyn::check_launch("dyn_sleep_us");
}
