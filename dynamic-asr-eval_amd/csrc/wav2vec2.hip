// Kernels specific to the wav2vec2 encoder (reference wav2vec2/lib.py:20-23: AutoModelForCTC = HF Wav2Vec2ForCTC;
// driven by dynamic_eval_ctc_loss :41-235 and dynamic_eval_ctc_loss_su :293-462).  Everything dense reuses
// dyn_gemm_f32 — the strided Conv1d feature extractor is an IMPLICIT GEMM over overlapping rows of the channels-last
// activation (row stride = stride * C_in < K = kernel * C_in, no im2col buffer).  This file holds the rest:
//   * exact (erf) GELU forward / backward                          (HF `hidden_act: gelu`)
//   * GroupNorm(num_groups == channels) over time = per-(batch, channel) normalisation, forward / backward
//   * col2im for the strided-conv input gradient (sums the overlapping rows)
//   * grouped positional conv (k = 128, 16 groups): pack / unpack between [B, T, C] and the group-major, zero-padded
//     [B, G, T + 2 pad, C/G] layout in which every group is again an overlapping-row GEMM
//   * weight normalisation of the positional conv (w = g * v / ||v||, norm over (out, in) per tap) forward / backward.
// All HBM-bound single passes with coalesced channel-contiguous accesses; reductions are fixed-order (deterministic).
#include "common.h"
#include "reduce.h"

namespace {
constexpr int TPB = 256;

inline unsigned grid_for(int64_t n) {
    int64_t g = dyn::cdiv(n, TPB);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

__global__ __launch_bounds__(TPB) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) y[i] = gelu_f(x[i]);
}
__global__ __launch_bounds__(TPB) void gelu_bwd_kernel(const float* __restrict__ x, const float* dy, float* dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) dx[i] = dy[i] * gelu_grad(x[i]);
}

// ---- per-(batch, channel) normalisation over time: x [B, T, C] ----------------------------------------------------
// stage 1: partial[(b, chunk), 0:C] = sum_t x, partial[(b, chunk), C:2C] = sum_t x^2 over the chunk's rows
// `valid` (device scalar, may be null): only the first *valid rows of every batch entry count (a zero-padded bucket, see dyn_colnorm_fwd_len)
__device__ __forceinline__ int64_t valid_rows(const int32_t* valid, int64_t T) {
    if (!valid) return T;
    const int64_t v = *valid;
    return v < 0 ? 0 : (v < T ? v : T);
}

__global__ __launch_bounds__(TPB) void colstats_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int64_t T,
                                                               int C, int64_t rows_per_chunk, int chunks, const int32_t* __restrict__ valid) {
    const int c = blockIdx.x * TPB + threadIdx.x;
    const int ch = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (c >= C) return;
    const int64_t Tv = valid_rows(valid, T);
    const int64_t r0 = (int64_t)ch * rows_per_chunk;
    const int64_t r1 = (r0 + rows_per_chunk < Tv) ? r0 + rows_per_chunk : Tv;
    float s = 0.f, q = 0.f;
    const float* xb = x + b * T * C + c;
    for (int64_t t = r0; t < r1; ++t) {
        const float v = xb[t * C];
        s += v;
        q += v * v;
    }
    float* p = partial + ((b * chunks + ch) * 2) * C;
    p[c] = s;
    p[C + c] = q;
}

// stage 2 (per batch): mean/rstd [B, C] from the chunk partials, summed in chunk order
__global__ __launch_bounds__(TPB) void colstats_final_kernel(const float* __restrict__ partial, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int64_t T, int C, int chunks, float eps,
                                                             const int32_t* __restrict__ valid) {
    const int c = blockIdx.x * TPB + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (c >= C) return;
    T = valid_rows(valid, T);
    if (T < 1) T = 1;
    double s = 0.0, q = 0.0;
    for (int ch = 0; ch < chunks; ++ch) {
        const float* p = partial + ((b * chunks + ch) * 2) * C;
        s += (double)p[c];
        q += (double)p[C + c];
    }
    const double m = s / (double)T;
    double var = q / (double)T - m * m;  // biased variance (GroupNorm / InstanceNorm)
    if (var < 0.0) var = 0.0;
    mean[b * C + c] = (float)m;
    rstd[b * C + c] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(TPB) void colnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y, int64_t B,
                                                            int64_t T, int C) {
    const int64_t total = B * T * C;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int c = (int)(i % C);
        const int64_t b = i / (T * C);
        y[i] = (x[i] - mean[b * C + c]) * rstd[b * C + c] * gamma[c] + beta[c];
    }
}

// backward stage 1: partial sums over time of dy and dy * xhat  (per batch, chunk)
__global__ __launch_bounds__(TPB) void colnorm_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  float* __restrict__ partial, int64_t T, int C,
                                                                  int64_t rows_per_chunk, int chunks, const int32_t* __restrict__ valid) {
    const int c = blockIdx.x * TPB + threadIdx.x;
    const int ch = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (c >= C) return;
    const int64_t Tv = valid_rows(valid, T);
    const int64_t r0 = (int64_t)ch * rows_per_chunk;
    const int64_t r1 = (r0 + rows_per_chunk < Tv) ? r0 + rows_per_chunk : Tv;
    const float m = mean[b * C + c], rs = rstd[b * C + c];
    float s1 = 0.f, s2 = 0.f;
    for (int64_t t = r0; t < r1; ++t) {
        const int64_t i = (b * T + t) * C + c;
        const float g = dy[i];
        s1 += g;
        s2 += g * (x[i] - m) * rs;
    }
    float* p = partial + ((b * chunks + ch) * 2) * C;
    p[c] = s1;
    p[C + c] = s2;
}

// backward stage 2: per-(b, c) sums -> sums [B, 2, C]; also dgamma / dbeta partial rows [B, 2C] for the ordered reducer
__global__ __launch_bounds__(TPB) void colnorm_bwd_sums_kernel(const float* __restrict__ partial, float* __restrict__ sums, int C,
                                                               int chunks) {
    const int c = blockIdx.x * TPB + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (c >= C) return;
    float s1 = 0.f, s2 = 0.f;
    for (int ch = 0; ch < chunks; ++ch) {
        const float* p = partial + ((b * chunks + ch) * 2) * C;
        s1 += p[c];
        s2 += p[C + c];
    }
    const int64_t Bn = gridDim.y;
    sums[(0 * Bn + b) * C + c] = s1;   // sum_t dy      -> dbeta contribution   (layout [2][B][C])
    sums[(1 * Bn + b) * C + c] = s2;   // sum_t dy*xhat -> dgamma contribution
}

// dx = rstd * gamma * (dy - mean_t(dy) - xhat * mean_t(dy * xhat))
__global__ __launch_bounds__(TPB) void colnorm_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ sums,
                                                                float* __restrict__ dx, int64_t B, int64_t T, int C,
                                                                const int32_t* __restrict__ valid) {
    const int64_t total = B * T * C;
    const int64_t Tv = valid_rows(valid, T);
    const float invT = 1.f / (float)(Tv > 0 ? Tv : 1);
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int c = (int)(i % C);
        const int64_t b = i / (T * C);
        if ((i / C) % T >= Tv) { dx[i] = 0.f; continue; }   // a padded row took no part in the statistics: no gradient
        const float rs = rstd[b * C + c];
        const float xh = (x[i] - mean[b * C + c]) * rs;
        const float s1 = sums[(0 * B + b) * C + c] * invT, s2 = sums[(1 * B + b) * C + c] * invT;
        dx[i] = rs * gamma[c] * (dy[i] - s1 - xh * s2);
    }
}

// ---- strided Conv1d input gradient: dx[b, r, ci] = sum_{j, t': s t' + j = r} dA[b, t', j * C + ci] ------------------
__global__ __launch_bounds__(TPB) void col2im_kernel(const float* __restrict__ dA, float* __restrict__ dx, int64_t B, int64_t Tin,
                                                     int64_t Tout, int C, int kw, int stride) {
    const int64_t total = B * Tin * C;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int ci = (int)(i % C);
        const int64_t r = (i / C) % Tin, b = i / (Tin * C);
        float s = 0.f;
        for (int j = 0; j < kw; ++j) {
            const int64_t rr = r - j;
            if (rr < 0 || rr % stride) continue;
            const int64_t tp = rr / stride;
            if (tp >= Tout) continue;
            s += dA[(b * Tout + tp) * (int64_t)(kw * C) + (int64_t)j * C + ci];
        }
        dx[i] = s;
    }
}

// x[b, t, :] = 0 for t >= *valid  (the padded tail of a bucket must look like the zero padding the positional conv sees past the last frame)
__global__ __launch_bounds__(TPB) void mask_rows_kernel(float* __restrict__ x, int64_t B, int64_t T, int C, const int32_t* __restrict__ valid) {
    const int64_t Tv = valid_rows(valid, T);
    const int64_t tail = (T - Tv) * C, total = B * tail;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB)
        x[(i / tail) * T * C + Tv * C + i % tail] = 0.f;
}

// ---- grouped positional conv: [B, T, C] <-> [B, G, T + 2 pad, C/G] (zero padded) ----------------------------------
__global__ __launch_bounds__(TPB) void group_pack_kernel(const float* __restrict__ x, float* __restrict__ xg, int64_t B, int64_t T,
                                                         int C, int G, int pad) {
    const int cg = C / G;
    const int64_t Tp = T + 2 * pad;
    const int64_t total = B * G * Tp * cg;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int c = (int)(i % cg);
        const int64_t tp = (i / cg) % Tp;
        const int g = (int)((i / (cg * Tp)) % G);
        const int64_t b = i / ((int64_t)cg * Tp * G);
        const int64_t t = tp - pad;
        xg[i] = (t >= 0 && t < T) ? x[(b * T + t) * C + g * cg + c] : 0.f;
    }
}

// y[b, t, g*cg + c] = yg[b, g, t, c]  (yg rows: Tg >= T; only the first T rows are used — HF drops the last frame)
__global__ __launch_bounds__(TPB) void group_unpack_kernel(const float* __restrict__ yg, float* __restrict__ y, int64_t B, int64_t T,
                                                           int64_t Tg, int C, int G, const float* __restrict__ bias) {
    const int cg = C / G;
    const int64_t total = B * T * C;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int ch = (int)(i % C);
        const int64_t t = (i / C) % T, b = i / (T * C);
        const int g = ch / cg, c = ch % cg;
        y[i] = yg[((b * G + g) * Tg + t) * cg + c] + (bias ? bias[ch] : 0.f);
    }
}

// dyg[b, g, t, c] = dy[b, t, g*cg + c] for t < T, 0 for T <= t < Tg
__global__ __launch_bounds__(TPB) void group_pack_grad_kernel(const float* __restrict__ dy, float* __restrict__ dyg, int64_t B,
                                                              int64_t T, int64_t Tg, int C, int G) {
    const int cg = C / G;
    const int64_t total = B * G * Tg * cg;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int c = (int)(i % cg);
        const int64_t t = (i / cg) % Tg;
        const int g = (int)((i / (cg * Tg)) % G);
        const int64_t b = i / ((int64_t)cg * Tg * G);
        dyg[i] = t < T ? dy[(b * T + t) * C + g * cg + c] : 0.f;
    }
}

// dx[b, t, g*cg + c] (+)= dxg[b, g, t + pad, c]
__global__ __launch_bounds__(TPB) void group_unpack_grad_kernel(const float* __restrict__ dxg, float* __restrict__ dx, int64_t B,
                                                                int64_t T, int C, int G, int pad, float beta) {
    const int cg = C / G;
    const int64_t Tp = T + 2 * pad;
    const int64_t total = B * T * C;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int ch = (int)(i % C);
        const int64_t t = (i / C) % T, b = i / (T * C);
        const int g = ch / cg, c = ch % cg;
        const float v = dxg[((b * G + g) * Tp + t + pad) * cg + c];
        dx[i] = beta != 0.f ? v + beta * dx[i] : v;
    }
}

// ---- weight norm over (out, in) per tap, packed layout v[G][co][kw][ci]: tap j of element e is (e / cg) % kw --------
// stage 1: partial[block, 0:kw] = sum v^2 per tap, partial[block, kw:2kw] = sum dw * v per tap, over this block's rows.
// A "row" is one (group, out-channel) slice of kw * cg contiguous elements; every thread owns fixed positions of the
// row pattern, so its running sums need no atomics; the cg positions of a tap are then added in order (deterministic).
constexpr int WN_MAX = 6144;  // kw * cg of the base model (128 * 48)
__global__ __launch_bounds__(TPB) void wn_partial_kernel(const float* __restrict__ v, const float* __restrict__ dw,
                                                         float* __restrict__ partial, int64_t rows, int kw, int cg,
                                                         int64_t rows_per_block) {
    __shared__ float pos[2][WN_MAX];
    const int rl = kw * cg;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
    for (int o = threadIdx.x; o < rl; o += TPB) {
        float a = 0.f, d = 0.f;
        for (int64_t r = r0; r < r1; ++r) {
            const float vv = v[r * rl + o];
            a += vv * vv;
            if (dw) d += dw[r * rl + o] * vv;
        }
        pos[0][o] = a;
        pos[1][o] = d;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < kw; j += TPB) {
        float a = 0.f, d = 0.f;
        for (int c = 0; c < cg; ++c) { a += pos[0][j * cg + c]; d += pos[1][j * cg + c]; }
        partial[(int64_t)blockIdx.x * 2 * kw + j] = a;
        partial[(int64_t)blockIdx.x * 2 * kw + kw + j] = d;
    }
}

// w = g[j] * v / ||v_j||
__global__ __launch_bounds__(TPB) void wn_apply_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                       const float* __restrict__ sums, float* __restrict__ w, int64_t n, int kw, int cg) {
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int j = (int)((e / cg) % kw);
        w[e] = g[j] * v[e] * rsqrtf(sums[j]);
    }
}

// dv = g/||v|| * (dw - v * <dw, v> / ||v||^2) ;  dg[j] (+)= <dw, v>_j / ||v_j||
__global__ __launch_bounds__(TPB) void wn_bwd_apply_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                           const float* __restrict__ dw, const float* __restrict__ sums,
                                                           float* __restrict__ dv, float* __restrict__ dg, float beta, int64_t n,
                                                           int kw, int cg) {
    for (int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x; e < n; e += (int64_t)gridDim.x * TPB) {
        const int j = (int)((e / cg) % kw);
        const float nn = sums[j], dot = sums[kw + j];
        const float inv = rsqrtf(nn);
        const float r = g[j] * inv * (dw[e] - v[e] * dot / nn);
        dv[e] = beta != 0.f ? r + beta * dv[e] : r;
    }
    if (blockIdx.x == 0)
        for (int j = threadIdx.x; j < kw; j += TPB) {
            const float r = sums[kw + j] * rsqrtf(sums[j]);
            dg[j] = beta != 0.f ? r + beta * dg[j] : r;
        }
}

inline int chunks_for(int64_t T, int64_t* per) {
    int64_t ch = dyn::cdiv(T, 64);
    if (ch > 256) ch = 256;
    if (ch < 1) ch = 1;
    *per = dyn::cdiv(T > 0 ? T : 1, ch);
    return (int)dyn::cdiv(T > 0 ? T : 1, *per);
}
}  // namespace

extern "C" int dyn_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
    DYN_REQUIRE(n >= 0 && (n == 0 || (x && y)), DYN_E_ARG, "dyn_gelu_fwd: bad arguments");
    if (n == 0) return DYN_OK;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, x, y, n);
    return dyn::check_launch("dyn_gelu_fwd");
}
extern "C" int dyn_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
    DYN_REQUIRE(n >= 0 && (n == 0 || (x && dy && dx)), DYN_E_ARG, "dyn_gelu_bwd: bad arguments");
    if (n == 0) return DYN_OK;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, x, dy, dx, n);
    return dyn::check_launch("dyn_gelu_bwd");
}

extern "C" int64_t dyn_colnorm_workspace_bytes(int64_t B, int64_t T, int64_t C) {
    int64_t per;
    return ((int64_t)B * chunks_for(T, &per) * 2 * C + (int64_t)B * 2 * C) * (int64_t)sizeof(float);
}

// `valid_rows`: null, or a DEVICE int32 scalar read when the kernels run — statistics over the first *valid_rows rows of every batch entry
// (rows past it are normalised with the same statistics: finite, meaningless).  One captured launch sequence serves every utterance length of
// a bucket this way (wav2vec2_model.py: the length lives in HBM, not in the launch arguments).
extern "C" int dyn_colnorm_fwd_len(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t B,
                                   int64_t T, int64_t C, float eps, const int32_t* valid_rows, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
    DYN_REQUIRE(x && gamma && beta && y && mean && rstd && B > 0 && T > 0 && C > 0, DYN_E_ARG, "dyn_colnorm_fwd: bad arguments");
    DYN_REQUIRE(workspace && workspace_bytes >= dyn_colnorm_workspace_bytes(B, T, C), DYN_E_WORKSPACE, "dyn_colnorm_fwd: workspace too small");
    int64_t per;
    const int chunks = chunks_for(T, &per);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(colstats_partial_kernel, dim3((unsigned)dyn::cdiv(C, TPB), (unsigned)chunks, (unsigned)B), dim3(TPB), 0, st, x, partial,
                       T, (int)C, per, chunks, valid_rows);
    hipLaunchKernelGGL(colstats_final_kernel, dim3((unsigned)dyn::cdiv(C, TPB), (unsigned)B), dim3(TPB), 0, st, partial, mean, rstd, T, (int)C,
                       chunks, eps, valid_rows);
    hipLaunchKernelGGL(colnorm_apply_kernel, dim3(grid_for(B * T * C)), dim3(TPB), 0, st, x, mean, rstd, gamma, beta, y, B, T, (int)C);
    return dyn::check_launch("dyn_colnorm_fwd");
}
extern "C" int dyn_colnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t B,
                               int64_t T, int64_t C, float eps, void* workspace, int64_t workspace_bytes, void* stream) {
    return dyn_colnorm_fwd_len(x, gamma, beta, y, mean, rstd, B, T, C, eps, nullptr, workspace, workspace_bytes, stream);
}

// `valid_rows` as in dyn_colnorm_fwd_len: sums over the valid rows only, dx = 0 on the rows past them.
extern "C" int dyn_colnorm_bwd_len(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy, float* dx,
                                   float* dgamma, float* dbeta, float wgrad_beta, int64_t B, int64_t T, int64_t C, const int32_t* valid_rows,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(x && gamma && mean && rstd && dy && dx && B > 0 && T > 0 && C > 0, DYN_E_ARG, "dyn_colnorm_bwd: bad arguments");
    DYN_REQUIRE(workspace && workspace_bytes >= dyn_colnorm_workspace_bytes(B, T, C), DYN_E_WORKSPACE, "dyn_colnorm_bwd: workspace too small");
    int64_t per;
    const int chunks = chunks_for(T, &per);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* sums = partial + (int64_t)B * chunks * 2 * C;  // [B, 2, C]
    hipLaunchKernelGGL(colnorm_bwd_partial_kernel, dim3((unsigned)dyn::cdiv(C, TPB), (unsigned)chunks, (unsigned)B), dim3(TPB), 0, st, x, dy,
                       mean, rstd, partial, T, (int)C, per, chunks, valid_rows);
    hipLaunchKernelGGL(colnorm_bwd_sums_kernel, dim3((unsigned)dyn::cdiv(C, TPB), (unsigned)B), dim3(TPB), 0, st, partial, sums, (int)C, chunks);
    hipLaunchKernelGGL(colnorm_bwd_apply_kernel, dim3(grid_for(B * T * C)), dim3(TPB), 0, st, x, dy, mean, rstd, gamma, sums, dx, B, T, (int)C,
                       valid_rows);
    // sums is [2][B][C]: reduce over the batch (in order) into the affine gradients
    dyn::ordered_before_launch(st);
    if (dbeta) dyn::launch_reduce_partials(sums, dbeta, B, C, wgrad_beta, st);
    if (dgamma) dyn::launch_reduce_partials(sums + (int64_t)B * C, dgamma, B, C, wgrad_beta, st);
    return dyn::check_launch("dyn_colnorm_bwd");
}
extern "C" int dyn_colnorm_bwd(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy, float* dx,
                               float* dgamma, float* dbeta, float wgrad_beta, int64_t B, int64_t T, int64_t C, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    return dyn_colnorm_bwd_len(x, gamma, mean, rstd, dy, dx, dgamma, dbeta, wgrad_beta, B, T, C, nullptr, workspace, workspace_bytes, stream);
}

// x [B, T, C]: rows t >= *valid_rows (device int32 scalar) of every batch entry are set to zero, the others are not touched.
extern "C" int dyn_mask_rows(float* x, int64_t B, int64_t T, int64_t C, const int32_t* valid_rows, void* stream) {
    DYN_REQUIRE(x && valid_rows && B > 0 && T > 0 && C > 0, DYN_E_ARG, "dyn_mask_rows: bad arguments");
    hipLaunchKernelGGL(mask_rows_kernel, dim3(grid_for(B * T * C)), dim3(TPB), 0, (hipStream_t)stream, x, B, T, (int)C, valid_rows);
    return dyn::check_launch("dyn_mask_rows");
}

extern "C" int64_t dyn_weight_norm_workspace_bytes(int64_t rows, int64_t kw) {
    int64_t nb = dyn::cdiv(rows, 8);
    if (nb > 256) nb = 256;
    return (nb * 2 * kw + 2 * kw) * (int64_t)sizeof(float);
}

// v, w: [rows][kw][cg] (rows = groups * out-channels-per-group); g: [kw].  w = g[j] * v / ||v[:, j, :]||_F
extern "C" int dyn_weight_norm_fwd(const float* v, const float* g, float* w, int64_t rows, int64_t kw, int64_t cg, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(v && g && w && rows > 0 && kw > 0 && cg > 0 && kw * cg <= WN_MAX, DYN_E_ARG, "dyn_weight_norm_fwd: bad arguments");
    DYN_REQUIRE(workspace && workspace_bytes >= dyn_weight_norm_workspace_bytes(rows, kw), DYN_E_WORKSPACE, "dyn_weight_norm_fwd: workspace too small");
    int64_t nb = dyn::cdiv(rows, 8);
    if (nb > 256) nb = 256;
    const int64_t per = dyn::cdiv(rows, nb);
    nb = dyn::cdiv(rows, per);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* sums = partial + nb * 2 * kw;
    hipLaunchKernelGGL(wn_partial_kernel, dim3((unsigned)nb), dim3(TPB), 0, st, v, (const float*)nullptr, partial, rows, (int)kw, (int)cg, per);
    dyn::launch_reduce_partials(partial, sums, nb, 2 * kw, 0.f, st);
    hipLaunchKernelGGL(wn_apply_kernel, dim3(grid_for(rows * kw * cg)), dim3(TPB), 0, st, v, g, sums, w, rows * kw * cg, (int)kw, (int)cg);
    return dyn::check_launch("dyn_weight_norm_fwd");
}

// dv (+)= d w / d v . dw ;  dg (+)= d w / d g . dw      (beta scales the existing dv / dg)
extern "C" int dyn_weight_norm_bwd(const float* v, const float* g, const float* dw, float* dv, float* dg, float beta, int64_t rows,
                                   int64_t kw, int64_t cg, void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(v && g && dw && dv && dg && rows > 0 && kw > 0 && cg > 0 && kw * cg <= WN_MAX, DYN_E_ARG, "dyn_weight_norm_bwd: bad arguments");
    DYN_REQUIRE(workspace && workspace_bytes >= dyn_weight_norm_workspace_bytes(rows, kw), DYN_E_WORKSPACE, "dyn_weight_norm_bwd: workspace too small");
    int64_t nb = dyn::cdiv(rows, 8);
    if (nb > 256) nb = 256;
    const int64_t per = dyn::cdiv(rows, nb);
    nb = dyn::cdiv(rows, per);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* sums = partial + nb * 2 * kw;
    hipLaunchKernelGGL(wn_partial_kernel, dim3((unsigned)nb), dim3(TPB), 0, st, v, dw, partial, rows, (int)kw, (int)cg, per);
    dyn::launch_reduce_partials(partial, sums, nb, 2 * kw, 0.f, st);
    hipLaunchKernelGGL(wn_bwd_apply_kernel, dim3(grid_for(rows * kw * cg)), dim3(TPB), 0, st, v, g, dw, sums, dv, dg, beta, rows * kw * cg,
                       (int)kw, (int)cg);
    return dyn::check_launch("dyn_weight_norm_bwd");
}

extern "C" int dyn_col2im_1d(const float* dA, float* dx, int64_t B, int64_t Tin, int64_t Tout, int64_t C, int64_t kw, int64_t stride,
                             void* stream) {
    DYN_REQUIRE(dA && dx && B > 0 && Tin > 0 && Tout > 0 && C > 0 && kw > 0 && stride > 0, DYN_E_ARG, "dyn_col2im_1d: bad arguments");
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(B * Tin * C)), dim3(TPB), 0, (hipStream_t)stream, dA, dx, B, Tin, Tout, (int)C, (int)kw,
                       (int)stride);
    return dyn::check_launch("dyn_col2im_1d");
}

extern "C" int dyn_group_pack(const float* x, float* xg, int64_t B, int64_t T, int64_t C, int64_t G, int64_t pad, void* stream) {
    DYN_REQUIRE(x && xg && B > 0 && T > 0 && C > 0 && G > 0 && C % G == 0 && pad >= 0, DYN_E_ARG, "dyn_group_pack: bad arguments");
    hipLaunchKernelGGL(group_pack_kernel, dim3(grid_for(B * (T + 2 * pad) * C)), dim3(TPB), 0, (hipStream_t)stream, x, xg, B, T, (int)C, (int)G,
                       (int)pad);
    return dyn::check_launch("dyn_group_pack");
}
extern "C" int dyn_group_unpack(const float* yg, float* y, const float* bias, int64_t B, int64_t T, int64_t Tg, int64_t C, int64_t G,
                                void* stream) {
    DYN_REQUIRE(yg && y && B > 0 && T > 0 && Tg >= T && C > 0 && G > 0 && C % G == 0, DYN_E_ARG, "dyn_group_unpack: bad arguments");
    hipLaunchKernelGGL(group_unpack_kernel, dim3(grid_for(B * T * C)), dim3(TPB), 0, (hipStream_t)stream, yg, y, B, T, Tg, (int)C, (int)G, bias);
    return dyn::check_launch("dyn_group_unpack");
}
extern "C" int dyn_group_pack_grad(const float* dy, float* dyg, int64_t B, int64_t T, int64_t Tg, int64_t C, int64_t G, void* stream) {
    DYN_REQUIRE(dy && dyg && B > 0 && T > 0 && Tg >= T && C > 0 && G > 0 && C % G == 0, DYN_E_ARG, "dyn_group_pack_grad: bad arguments");
    hipLaunchKernelGGL(group_pack_grad_kernel, dim3(grid_for(B * Tg * C)), dim3(TPB), 0, (hipStream_t)stream, dy, dyg, B, T, Tg, (int)C, (int)G);
    return dyn::check_launch("dyn_group_pack_grad");
}
extern "C" int dyn_group_unpack_grad(const float* dxg, float* dx, int64_t B, int64_t T, int64_t C, int64_t G, int64_t pad, float beta,
                                     void* stream) {
    DYN_REQUIRE(dxg && dx && B > 0 && T > 0 && C > 0 && G > 0 && C % G == 0, DYN_E_ARG, "dyn_group_unpack_grad: bad arguments");
    hipLaunchKernelGGL(group_unpack_grad_kernel, dim3(grid_for(B * T * C)), dim3(TPB), 0, (hipStream_t)stream, dxg, dx, B, T, (int)C, (int)G,
                       (int)pad, beta);
    return dyn::check_launch("dyn_group_unpack_grad");
}
