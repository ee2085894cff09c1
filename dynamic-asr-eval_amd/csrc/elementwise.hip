// HBM-bound elementwise pieces of the encoder forward/backward (16-B vector accesses, grid-stride):
// SiLU / ReLU(not used) / GLU, residual axpy, column sums for bias gradients, and the deterministic
// partial-sum reducer shared by every weight-gradient reduction.
// Reference call sites: the activation / residual / bias ops inside model(audio_signal=...) and
// loss.backward() (reference lcasr/lib.py:550,579); SpecAugment frequency masks (lib.py:541).
#include "common.h"
#include "reduce.h"

namespace {

constexpr int TPB = 256;

inline unsigned grid_for(int64_t n_vec) {
    int64_t g = dyn::cdiv(n_vec, TPB);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__device__ __forceinline__ float silu_f(float x) { return x * dyn::sigmoidf_(x); }
__device__ __forceinline__ float silu_grad(float x) {
    const float s = dyn::sigmoidf_(x);
    return s * (1.f + x * (1.f - s));
}

__global__ void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
        reinterpret_cast<float4*>(y)[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
        y[i] = silu_f(x[i]);
}

// dx = dy * silu'(x)   (dx may alias dy)
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* dy, float* dx, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        g.x *= silu_grad(v.x); g.y *= silu_grad(v.y); g.z *= silu_grad(v.z); g.w *= silu_grad(v.w);
        reinterpret_cast<float4*>(dx)[i] = g;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
        dx[i] = dy[i] * silu_grad(x[i]);
}

// GLU over the last dim: u [rows, 2C] -> y [rows, C] = u[:, :C] * sigmoid(u[:, C:]).   C % 4 == 0.
__global__ void glu_fwd_kernel(const float* __restrict__ u, float* __restrict__ y, int64_t rows, int C) {
    const int c4n = C >> 2;
    const int64_t total = rows * c4n;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int64_t r = i / c4n;
        const int c4 = (int)(i % c4n);
        const float4 a = reinterpret_cast<const float4*>(u + r * 2 * C)[c4];
        const float4 b = reinterpret_cast<const float4*>(u + r * 2 * C + C)[c4];
        float4 o;
        o.x = a.x * dyn::sigmoidf_(b.x); o.y = a.y * dyn::sigmoidf_(b.y);
        o.z = a.z * dyn::sigmoidf_(b.z); o.w = a.w * dyn::sigmoidf_(b.w);
        reinterpret_cast<float4*>(y + r * C)[c4] = o;
    }
}

__global__ void glu_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dy, float* __restrict__ du,
                               int64_t rows, int C) {
    const int c4n = C >> 2;
    const int64_t total = rows * c4n;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int64_t r = i / c4n;
        const int c4 = (int)(i % c4n);
        const float4 a = reinterpret_cast<const float4*>(u + r * 2 * C)[c4];
        const float4 b = reinterpret_cast<const float4*>(u + r * 2 * C + C)[c4];
        const float4 g = reinterpret_cast<const float4*>(dy + r * C)[c4];
        float4 da, db;
        float s;
        s = dyn::sigmoidf_(b.x); da.x = g.x * s; db.x = g.x * a.x * s * (1.f - s);
        s = dyn::sigmoidf_(b.y); da.y = g.y * s; db.y = g.y * a.y * s * (1.f - s);
        s = dyn::sigmoidf_(b.z); da.z = g.z * s; db.z = g.z * a.z * s * (1.f - s);
        s = dyn::sigmoidf_(b.w); da.w = g.w * s; db.w = g.w * a.w * s * (1.f - s);
        reinterpret_cast<float4*>(du + r * 2 * C)[c4] = da;
        reinterpret_cast<float4*>(du + r * 2 * C + C)[c4] = db;
    }
}

// y = a * x + b * y
__global__ void axpby_kernel(const float* __restrict__ x, float* y, float a, float b, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 w = reinterpret_cast<float4*>(y)[i];
        w.x = a * v.x + b * w.x; w.y = a * v.y + b * w.y; w.z = a * v.z + b * w.z; w.w = a * v.w + b * w.w;
        reinterpret_cast<float4*>(y)[i] = w;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
        y[i] = a * x[i] + b * y[i];
}

// Column sums, stage 1: x [rows, C] -> partial [nchunk, C]; each block owns a 64-column strip and a row chunk,
// 4 row-lanes x 64 column-lanes, combined through LDS in a fixed order (deterministic).
__global__ void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int64_t rows, int C,
                                      int64_t rows_per_chunk) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = (r0 + rows_per_chunk < rows) ? r0 + rows_per_chunk : rows;
    float s = 0.f;
    if (col < C)
        for (int64_t r = r0 + rl; r < r1; r += 4) s += x[r * C + col];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && col < C) partial[(int64_t)blockIdx.y * C + col] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// Column sums, 16 B per lane: a wave covers a 256-column strip of a row with one 1-KiB transaction; the four waves of a
// workgroup split the rows of a chunk and are combined in wave order through LDS; chunk partials go to the fixed-order
// 2-D reducer.  Thousands of waves instead of C/32 workgroups: bias gradients of [2048..8192, 768..4096] activations.
__global__ __launch_bounds__(256) void colsum_v4_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int64_t rows,
                                                                int C, int64_t rows_per_chunk) {
    __shared__ float4 red[3][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * 4;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t r1 = (r0 + rows_per_chunk < rows) ? r0 + rows_per_chunk : rows;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 < C)
        for (int64_t r = r0 + wv; r < r1; r += 4) {
            const float4 v = *reinterpret_cast<const float4*>(x + r * C + c0);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    if (wv > 0) red[wv - 1][lane] = s;
    __syncthreads();
    if (wv == 0 && c0 < C) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float4 v = red[k][lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.y * C + c0) = s;
    }
}

// Column sums in ONE launch for the common small case (bias gradients of a 2048..8192-row activation): 1024 threads =
// 32 columns x 32 row-lanes, one 128-B segment per row per wave half; lanes are combined through LDS in lane order.
__global__ __launch_bounds__(1024) void colsum_single_kernel(const float* __restrict__ x, float* out, int64_t rows, int C, float beta) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (col < C)
        for (int64_t r = rl; r < rows; r += 32) s += x[r * C + col];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && col < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][cl];
        out[col] = (beta != 0.f ? beta * out[col] : 0.f) + t;
    }
}

// out[n] = beta * out[n] + sum_p partial[p, n]   (p in increasing order)
__global__ void reduce_partials_kernel(const float* __restrict__ partial, float* out, int64_t P, int64_t n, float beta) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        float s = 0.f;
        for (int64_t p = 0; p < P; ++p) s += partial[p * n + i];
        out[i] = (beta != 0.f ? beta * out[i] : 0.f) + s;
    }
}

// SpecAugment frequency masking on a [F, T] log-mel window: rows f0[k] <= f < f0[k]+w[k] set to `value`.
__global__ void freq_mask_kernel(float* x, int F, int64_t T, const int32_t* __restrict__ f0, const int32_t* __restrict__ w,
                                 int n_masks, float value, const float* __restrict__ value_dev) {
    if (value_dev) value = *value_dev;
    const int f = blockIdx.y;
    bool hit = false;
    for (int k = 0; k < n_masks; ++k) hit |= (f >= f0[k] && f < f0[k] + w[k]);
    if (!hit) return;
    for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < T; t += (int64_t)gridDim.x * TPB) x[(int64_t)f * T + t] = value;
}

// SpecAugment time masking: columns t0[k] <= t < t0[k]+w[k] of every row set to `value`.
__global__ void time_mask_kernel(float* x, int F, int64_t T, const int32_t* __restrict__ t0, const int32_t* __restrict__ w,
                                 int n_masks, float value, const float* __restrict__ value_dev) {
    if (value_dev) value = *value_dev;
    const int k = blockIdx.y;
    const int64_t a = t0[k], wd = w[k];
    const int64_t total = (int64_t)F * wd;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int64_t f = i / wd, t = a + i % wd;
        if (t >= 0 && t < T) x[f * T + t] = value;
    }
}

// [F, T] (row-major, T contiguous) -> [T, F]: the log-mel window enters the encoder channels-last.
__global__ void transpose_ft_kernel(const float* __restrict__ x, float* __restrict__ y, int F, int64_t T, int64_t ldx) {
    __shared__ float tile[32][33];
    const int64_t t0 = (int64_t)blockIdx.x * 32;
    const int f0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int f = f0 + j;
        const int64_t t = t0 + tx;
        tile[j][tx] = (f < F && t < T) ? x[(int64_t)f * ldx + t] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t t = t0 + j;
        const int f = f0 + tx;
        if (t < T && f < F) y[t * F + f] = tile[tx][j];
    }
}

}  // namespace

extern "C" int dyn_silu_fwd(const float* x, float* y, int64_t n, void* stream) {
    DYN_REQUIRE(n >= 0 && (n == 0 || (x && y)), DYN_E_ARG, "dyn_silu_fwd: bad arguments");
    if (n == 0) return DYN_OK;
    hipLaunchKernelGGL(silu_fwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, (hipStream_t)stream, x, y, n);
    return dyn::check_launch("dyn_silu_fwd");
}

extern "C" int dyn_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
    DYN_REQUIRE(n >= 0 && (n == 0 || (x && dy && dx)), DYN_E_ARG, "dyn_silu_bwd: bad arguments");
    if (n == 0) return DYN_OK;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, (hipStream_t)stream, x, dy, dx, n);
    return dyn::check_launch("dyn_silu_bwd");
}

extern "C" int dyn_glu_fwd(const float* u, float* y, int64_t rows, int64_t C, void* stream) {
    DYN_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0 && u && y, DYN_E_ARG, "dyn_glu_fwd: need C %% 4 == 0 (C=%lld)", (long long)C);
    if (rows == 0) return DYN_OK;
    hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid_for(rows * C / 4)), dim3(TPB), 0, (hipStream_t)stream, u, y, rows, (int)C);
    return dyn::check_launch("dyn_glu_fwd");
}

extern "C" int dyn_glu_bwd(const float* u, const float* dy, float* du, int64_t rows, int64_t C, void* stream) {
    DYN_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0 && u && dy && du, DYN_E_ARG, "dyn_glu_bwd: need C %% 4 == 0");
    if (rows == 0) return DYN_OK;
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_for(rows * C / 4)), dim3(TPB), 0, (hipStream_t)stream, u, dy, du, rows, (int)C);
    return dyn::check_launch("dyn_glu_bwd");
}

extern "C" int dyn_axpby(const float* x, float* y, float a, float b, int64_t n, void* stream) {
    DYN_REQUIRE(n >= 0 && (n == 0 || (x && y)), DYN_E_ARG, "dyn_axpby: bad arguments");
    if (n == 0) return DYN_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, (hipStream_t)stream, x, y, a, b, n);
    return dyn::check_launch("dyn_axpby");
}

extern "C" int64_t dyn_colsum_workspace_bytes(int64_t rows, int64_t C) {
    int64_t chunks = dyn::cdiv(rows, 32);
    if (chunks > 512) chunks = 512;
    if (chunks < 1) chunks = 1;
    return chunks * C * (int64_t)sizeof(float);
}

extern "C" int dyn_colsum(const float* x, float* out, int64_t rows, int64_t C, float beta, void* workspace,
                          int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(rows >= 0 && C > 0 && x && out, DYN_E_ARG, "dyn_colsum: bad arguments");
    int64_t chunks = dyn::cdiv(rows, 32);
    if (chunks > 512) chunks = 512;
    if (chunks < 1) chunks = 1;
    DYN_REQUIRE(workspace && workspace_bytes >= chunks * C * (int64_t)sizeof(float), DYN_E_WORKSPACE,
                "dyn_colsum: workspace too small");
    const int64_t rpc = dyn::cdiv(rows > 0 ? rows : 1, chunks);
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)workspace) & 15) == 0 && rows >= 256) {
        const int64_t ch = dyn::cdiv(rows, rpc);
        float* part = dyn::partials_alloc(workspace, ch * C * (int64_t)sizeof(float));   // the workspace, or the open deferral context's arena
        hipLaunchKernelGGL(colsum_v4_partial_kernel, dim3((unsigned)dyn::cdiv(C, 256), (unsigned)ch), dim3(256), 0, st, x, part, rows, (int)C, rpc);
        dyn::reduce_or_defer(part, out, ch, C, beta, st);
        return dyn::check_launch("dyn_colsum");
    }
    if (rows <= 8192 && C >= 256) {  // enough column strips to occupy the chip: one launch instead of two
        hipLaunchKernelGGL(colsum_single_kernel, dim3((unsigned)dyn::cdiv(C, 32)), dim3(1024), 0, st, x, out, rows, (int)C, beta);
        return dyn::check_launch("dyn_colsum");
    }
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)dyn::cdiv(C, 64), (unsigned)chunks), dim3(256), 0, st, x,
                       (float*)workspace, rows, (int)C, rpc);
    dyn::ordered_before_launch(st);
    dyn::launch_reduce_partials((const float*)workspace, out, chunks, C, beta, st);
    return dyn::check_launch("dyn_colsum");
}

extern "C" int dyn_reduce_partials(const float* partial, float* out, int64_t P, int64_t n, float beta, void* stream) {
    DYN_REQUIRE(P >= 0 && n >= 0 && (n == 0 || (partial && out)), DYN_E_ARG, "dyn_reduce_partials: bad arguments");
    if (n == 0) return DYN_OK;
    dyn::ordered_before_launch((hipStream_t)stream);   // recorded reductions (dyn_reduce_defer_begin) may target the same output: they go first
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, partial, out, P, n, beta);
    return dyn::check_launch("dyn_reduce_partials");
}

extern "C" int dyn_specaug_freqmask(float* x, int64_t F, int64_t T, const int32_t* f0, const int32_t* width,
                                    int64_t n_masks, float value, const float* value_dev, void* stream) {
    DYN_REQUIRE(x && F > 0 && T >= 0 && n_masks >= 0 && (n_masks == 0 || (f0 && width)), DYN_E_ARG,
                "dyn_specaug_freqmask: bad arguments");
    if (T == 0 || n_masks == 0) return DYN_OK;
    int64_t gx = dyn::cdiv(T, TPB * 4);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(freq_mask_kernel, dim3((unsigned)gx, (unsigned)F), dim3(TPB), 0, (hipStream_t)stream, x, (int)F, T,
                       f0, width, (int)n_masks, value, value_dev);
    return dyn::check_launch("dyn_specaug_freqmask");
}

extern "C" int dyn_specaug_timemask(float* x, int64_t F, int64_t T, const int32_t* t0, const int32_t* width, int64_t n_masks,
                                    float value, const float* value_dev, void* stream) {
    DYN_REQUIRE(x && F > 0 && T >= 0 && n_masks >= 0 && (n_masks == 0 || (t0 && width)), DYN_E_ARG,
                "dyn_specaug_timemask: bad arguments");
    if (T == 0 || n_masks == 0) return DYN_OK;
    hipLaunchKernelGGL(time_mask_kernel, dim3(64, (unsigned)n_masks), dim3(TPB), 0, (hipStream_t)stream, x, (int)F, T, t0, width,
                       (int)n_masks, value, value_dev);
    return dyn::check_launch("dyn_specaug_timemask");
}

extern "C" int dyn_transpose_ft(const float* x, float* y, int64_t F, int64_t T, int64_t ldx, void* stream) {
    DYN_REQUIRE(x && y && F > 0 && T >= 0 && ldx >= T, DYN_E_ARG, "dyn_transpose_ft: bad arguments");
    if (T == 0) return DYN_OK;
    hipLaunchKernelGGL(transpose_ft_kernel, dim3((unsigned)dyn::cdiv(T, 32), (unsigned)dyn::cdiv(F, 32)), dim3(256), 0,
                       (hipStream_t)stream, x, y, (int)F, T, ldx);
    return dyn::check_launch("dyn_transpose_ft");
}
