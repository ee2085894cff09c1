// Shared helpers for the libdyneval_hip C-ABI (gfx950 / CDNA4 only).
// Every entry point returns 0 on success or a negative DYN_E_* code and records a
// thread-local message readable through dyn_last_error(); no exception crosses the ABI,
// no entry point allocates device memory or synchronises the stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dyneval.h"

namespace dyn {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return DYN_E_LAUNCH;
    }
    return DYN_OK;
}

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x a multiple of 64 (<= 1024); `red` holds >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// Counter-based randomness (include/dyneval.h, dyn_dropout): a draw is a pure function of (seed, stream, index).
__device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t stream, uint64_t index) {
    uint64_t z = seed ^ ((stream + 1) * 0x9E3779B97F4A7C15ull) ^ ((index + 1) * 0xC2B2AE3D27D4EB4Full);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// the Gumbel-max key of class c: x * inv_t - log(-log(u)), u = odd / 2^24 in (0, 1)
__device__ __forceinline__ float gumbel_key(float x, float inv_t, uint64_t seed, uint64_t stream, uint64_t c) {
    const float u = (float)(2 * (mix64(seed, stream, c) >> 41) + 1) * 5.9604644775390625e-8f;
    return x * inv_t - logf(-logf(u));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Deferred column reductions (dyn_reduce_defer_begin / dyn_reduce_defer_flush, include/dyneval.h).  A kernel that leaves per-workgroup
// partial sums takes their buffer from partials_alloc(): the caller's workspace as ever, or — while a deferral context is open on this
// thread and has room — a slice of the context's arena, which nothing else touches before the flush.  reduce_or_defer() then launches the
// reduction at once (partials in the workspace) or records it (partials in the arena).
float* partials_alloc(void* workspace, int64_t bytes);
void reduce_or_defer(const float* partial, float* out, int64_t P, int64_t n, float beta, hipStream_t st);
void reduce_taps_or_defer(const float* partial, float* out, int64_t P, int64_t n, float beta, int C, hipStream_t st);   // [tap][C] -> [C][taps]
void reduce_pair_or_defer(const float* p0, float* out0, const float* p1, float* out1, int64_t P, int64_t n, float beta, hipStream_t st);
// Every OTHER launch that writes a reduction output while a deferral context is open goes through ordered_before_launch() first: recording
// order is execution order, so what has been recorded so far runs before the direct launch (a recorded item into the same output must not be
// overtaken — today every writer uses beta = 1 and only the summation order would change, a beta = 0 writer would lose a term).
void ordered_before_launch(hipStream_t st);

}  // namespace dyn

#define DYN_REQUIRE(cond, code, ...)                         \
    do {                                                     \
        if (!(cond)) {                                       \
            dyn::set_error(__VA_ARGS__);                     \
            return (code);                                   \
        }                                                    \
    } while (0)
