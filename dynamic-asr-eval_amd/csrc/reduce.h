// Deterministic column reduction of partial results: out[n] = beta * out[n] + sum_p partial[p, n].
// 1024-thread workgroup = 64 columns x 16 row-lanes: each row-lane adds every 16th partial row (coalesced 256-B
// segments per wave), the 16 lane sums are combined through LDS in lane order.  Fixed order => bitwise reproducible.
#pragma once
#include "common.h"

namespace dyn {

__global__ __launch_bounds__(1024) static void reduce_partials_2d_kernel(const float* __restrict__ partial, float* out,
                                                                         int64_t P, int64_t n, float beta) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + cl;
    float s = 0.f;
    if (col < n)
        for (int64_t p = rl; p < P; p += 16) s += partial[p * n + col];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && col < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        out[col] = (beta != 0.f ? beta * out[col] : 0.f) + t;
    }
}

// Same reduction for partial rows laid out [j][C] (j = filter tap) into an output laid out [C][J]: the coalesced partial
// layout of the vectorised conv weight-gradient kernels.  out[c * J + j] = beta * out[...] + sum_p partial[p, j * C + c].
__global__ __launch_bounds__(1024) static void reduce_partials_2d_taps_kernel(const float* __restrict__ partial, float* out,
                                                                              int64_t P, int64_t n, float beta, int C) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + cl;
    float s = 0.f;
    if (col < n)
        for (int64_t p = rl; p < P; p += 16) s += partial[p * n + col];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && col < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        const int64_t J = n / C, o = (col % C) * J + col / C;
        out[o] = (beta != 0.f ? beta * out[o] : 0.f) + t;
    }
}

static inline void launch_reduce_partials_taps(const float* partial, float* out, int64_t P, int64_t n, float beta, int C, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_2d_taps_kernel, dim3((unsigned)cdiv(n, 64)), dim3(1024), 0, st, partial, out, P, n, beta, C);
}

// Two independent reductions of equal shape in one launch (blockIdx.y selects the pair): LayerNorm dgamma + dbeta.
__global__ __launch_bounds__(1024) static void reduce_partials_2d_pair_kernel(const float* __restrict__ p0, float* out0,
                                                                              const float* __restrict__ p1, float* out1,
                                                                              int64_t P, int64_t n, float beta) {
    __shared__ float red[16][64];
    const float* __restrict__ partial = blockIdx.y ? p1 : p0;
    float* out = blockIdx.y ? out1 : out0;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + cl;
    float s = 0.f;
    if (col < n)
        for (int64_t p = rl; p < P; p += 16) s += partial[p * n + col];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && col < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        out[col] = (beta != 0.f ? beta * out[col] : 0.f) + t;
    }
}

static inline void launch_reduce_partials_pair(const float* p0, float* out0, const float* p1, float* out1, int64_t P, int64_t n,
                                               float beta, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_2d_pair_kernel, dim3((unsigned)cdiv(n, 64), 2), dim3(1024), 0, st, p0, out0, p1, out1, P, n, beta);
}

static inline void launch_reduce_partials(const float* partial, float* out, int64_t P, int64_t n, float beta, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_2d_kernel, dim3((unsigned)cdiv(n, 64)), dim3(1024), 0, st, partial, out, P, n, beta);
}

}  // namespace dyn
