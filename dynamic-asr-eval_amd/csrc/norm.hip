// LayerNorm / RMSNorm forward + backward over the channel dim (rows x C, C % 256 == 0, C <= 2048).
// One wave64 per row: 16-B loads, the row lives in registers, mean / variance / gradient dot-products are
// wavefront reductions (no LDS, no barrier on the forward path).  Weight gradients are accumulated per
// workgroup in registers, written as partial rows and summed in a fixed order (deterministic).
// Reference call sites: every LayerNorm / conv-module norm inside model(audio_signal=...) and its backward
// (reference lcasr/lib.py:550,579); `default_norm: layer_norm` (earnings_finetune/lcasr160rb1.yaml:24).
#include "common.h"
#include "reduce.h"

namespace {

constexpr int WPB = 4;  // waves (rows in flight) per workgroup

// NV = C / 256 float4 per lane.
template <int NV, bool RMS>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                        int64_t rows, float eps, int64_t rps, int ngroups, int64_t pstride) {
    // Lockstep groups (several recordings, each with its own adapted weights, in one batch): blockIdx.y = sample, a sample owns `rps`
    // consecutive rows and takes the parameters of group (sample % ngroups) at gamma + group * pstride.  Ungrouped: gridDim.y = 1, rps = rows.
    constexpr int C = NV * 256;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row_lo = (int64_t)blockIdx.y * rps;
    const int64_t row_hi = row_lo + rps < rows ? row_lo + rps : rows;
    const int64_t poff = (int64_t)(blockIdx.y % ngroups) * pstride;
    gamma += poff;
    if (!RMS && beta) beta += poff;
    float4 g[NV], b[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        g[j] = reinterpret_cast<const float4*>(gamma)[lane + 64 * j];
        b[j] = (!RMS && beta) ? reinterpret_cast<const float4*>(beta)[lane + 64 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t row = row_lo + (int64_t)blockIdx.x * WPB + w; row < row_hi; row += (int64_t)gridDim.x * WPB) {
        float4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = reinterpret_cast<const float4*>(x + row * C)[lane + 64 * j];
            s += RMS ? (v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w)
                     : (v[j].x + v[j].y + v[j].z + v[j].w);
        }
        s = dyn::wave_sum(s);
        float mean = 0.f, var;
        if (RMS) {
            var = s / C;
        } else {
            mean = s / C;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const float a0 = v[j].x - mean, a1 = v[j].y - mean, a2 = v[j].z - mean, a3 = v[j].w - mean;
                q += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
            }
            var = dyn::wave_sum(q) / C;
        }
        const float rstd = rsqrtf(var + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 o;
            o.x = (v[j].x - mean) * rstd * g[j].x + b[j].x;
            o.y = (v[j].y - mean) * rstd * g[j].y + b[j].y;
            o.z = (v[j].z - mean) * rstd * g[j].z + b[j].z;
            o.w = (v[j].w - mean) * rstd * g[j].w + b[j].w;
            reinterpret_cast<float4*>(y + row * C)[lane + 64 * j] = o;
        }
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
    }
}

// dx (+)= norm_bwd(dy);  partial_g/partial_b [gridDim.x, C] get this workgroup's weight-gradient sums.
template <int NV, bool RMS>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                        const float* __restrict__ dy, float* dx, float dx_beta,
                                                        float* __restrict__ partial_g, float* __restrict__ partial_b,
                                                        int64_t rows, const float* dx_in, int64_t rps, int ngroups, int64_t pstride) {
    constexpr int C = NV * 256;
    __shared__ float4 red[WPB][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row_lo = (int64_t)blockIdx.y * rps;                      // see norm_fwd_kernel: blockIdx.y = sample of a lockstep group
    const int64_t row_hi = row_lo + rps < rows ? row_lo + rps : rows;
    gamma += (int64_t)(blockIdx.y % ngroups) * pstride;
    float4 g[NV], ag[NV], ab[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        g[j] = reinterpret_cast<const float4*>(gamma)[lane + 64 * j];
        ag[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t row = row_lo + (int64_t)blockIdx.x * WPB + w; row < row_hi; row += (int64_t)gridDim.x * WPB) {
        const float mean = RMS ? 0.f : mean_in[row];
        const float rstd = rstd_in[row];
        float4 xh[NV], gy[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 v = reinterpret_cast<const float4*>(x + row * C)[lane + 64 * j];
            const float4 d = reinterpret_cast<const float4*>(dy + row * C)[lane + 64 * j];
            xh[j].x = (v.x - mean) * rstd; xh[j].y = (v.y - mean) * rstd;
            xh[j].z = (v.z - mean) * rstd; xh[j].w = (v.w - mean) * rstd;
            ag[j].x += d.x * xh[j].x; ag[j].y += d.y * xh[j].y; ag[j].z += d.z * xh[j].z; ag[j].w += d.w * xh[j].w;
            ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
            gy[j].x = d.x * g[j].x; gy[j].y = d.y * g[j].y; gy[j].z = d.z * g[j].z; gy[j].w = d.w * g[j].w;
            s1 += gy[j].x + gy[j].y + gy[j].z + gy[j].w;
            s2 += gy[j].x * xh[j].x + gy[j].y * xh[j].y + gy[j].z * xh[j].z + gy[j].w * xh[j].w;
        }
        s2 = dyn::wave_sum(s2) / C;
        s1 = RMS ? 0.f : dyn::wave_sum(s1) / C;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 o;
            o.x = rstd * (gy[j].x - s1 - xh[j].x * s2);
            o.y = rstd * (gy[j].y - s1 - xh[j].y * s2);
            o.z = rstd * (gy[j].z - s1 - xh[j].z * s2);
            o.w = rstd * (gy[j].w - s1 - xh[j].w * s2);
            float4* p = reinterpret_cast<float4*>(dx + row * C) + lane + 64 * j;
            if (dx_beta != 0.f) {
                const float4 old = reinterpret_cast<const float4*>(dx_in + row * C)[lane + 64 * j];   // dx_in == dx: in place
                o.x += dx_beta * old.x; o.y += dx_beta * old.y; o.z += dx_beta * old.z; o.w += dx_beta * old.w;
            }
            *p = o;
        }
    }
    // Combine the 4 waves' weight-gradient sums in wave order, then write this workgroup's partial row.
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        for (int pass = 0; pass < (RMS ? 1 : 2); ++pass) {
            __syncthreads();
            red[w][lane] = pass == 0 ? ag[j] : ab[j];
            __syncthreads();
            if (w == 0) {
                float4 t = red[0][lane];
#pragma unroll
                for (int k = 1; k < WPB; ++k) { t.x += red[k][lane].x; t.y += red[k][lane].y; t.z += red[k][lane].z; t.w += red[k][lane].w; }
                float* dst = (pass == 0 ? partial_g : partial_b) + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * C;
                reinterpret_cast<float4*>(dst)[lane + 64 * j] = t;
            }
        }
    }
}

inline int bwd_blocks(int64_t rows) {
    int64_t g = dyn::cdiv(rows, WPB);
    if (g > 256) g = 256;
    if (g < 1) g = 1;
    return (int)g;
}

struct Grouping { int64_t rps; int n; int64_t pstride; };   // rows per sample, groups, parameter stride (elements); n <= 1: ungrouped

template <bool RMS>
int launch_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows,
               int64_t C, float eps, hipStream_t st, Grouping gr = {0, 1, 0}) {
    const bool grouped = gr.n > 1;
    const int64_t rps = grouped ? gr.rps : rows;
    const int ngroups = grouped ? gr.n : 1;
    const int64_t pstride = grouped ? gr.pstride : 0;
    int64_t gq = dyn::cdiv(rps, WPB);
    if (gq > 2048) gq = 2048;
    dim3 grid((unsigned)gq, grouped ? (unsigned)dyn::cdiv(rows, rps) : 1u), blk(256);
    switch (C / 256) {
        case 1: hipLaunchKernelGGL((norm_fwd_kernel<1, RMS>), grid, blk, 0, st, x, gamma, beta, y, mean, rstd, rows, eps, rps, ngroups, pstride); break;
        case 2: hipLaunchKernelGGL((norm_fwd_kernel<2, RMS>), grid, blk, 0, st, x, gamma, beta, y, mean, rstd, rows, eps, rps, ngroups, pstride); break;
        case 3: hipLaunchKernelGGL((norm_fwd_kernel<3, RMS>), grid, blk, 0, st, x, gamma, beta, y, mean, rstd, rows, eps, rps, ngroups, pstride); break;
        case 4: hipLaunchKernelGGL((norm_fwd_kernel<4, RMS>), grid, blk, 0, st, x, gamma, beta, y, mean, rstd, rows, eps, rps, ngroups, pstride); break;
        case 8: hipLaunchKernelGGL((norm_fwd_kernel<8, RMS>), grid, blk, 0, st, x, gamma, beta, y, mean, rstd, rows, eps, rps, ngroups, pstride); break;
        default: dyn::set_error("norm: unsupported C=%lld (need C in {256,512,768,1024,2048})", (long long)C); return DYN_E_UNSUPPORTED;
    }
    return dyn::check_launch("dyn_norm_fwd");
}

template <bool RMS>
int launch_bwd(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy, float* dx,
               float dx_beta, float* dgamma, float* dbeta, float wbeta, int64_t rows, int64_t C, void* ws, int64_t ws_bytes,
               hipStream_t st, const float* dx_in = nullptr, Grouping gr = {0, 1, 0}) {
    if (dx_in == nullptr) dx_in = dx;
    const bool grouped = gr.n > 1;
    const int64_t rps = grouped ? gr.rps : rows;
    const int ngroups = grouped ? gr.n : 1;
    const int64_t pstride = grouped ? gr.pstride : 0;
    const int nsamp = grouped ? (int)dyn::cdiv(rows, rps) : 1;
    const int nbs = bwd_blocks(rps);            // workgroups (= partial rows) per sample
    const int nb = nbs * nsamp;
    DYN_REQUIRE(ws && ws_bytes >= (int64_t)2 * nb * C * (int64_t)sizeof(float), DYN_E_WORKSPACE, "norm_bwd: workspace too small");
    float* pg = dyn::partials_alloc(ws, (int64_t)2 * nb * C * (int64_t)sizeof(float));   // the workspace, or the open deferral context's arena
    float* pb = pg + (int64_t)nb * C;
    dim3 grid(nbs, nsamp), blk(256);
    switch (C / 256) {
        case 1: hipLaunchKernelGGL((norm_bwd_kernel<1, RMS>), grid, blk, 0, st, x, gamma, mean, rstd, dy, dx, dx_beta, pg, pb, rows, dx_in, rps, ngroups, pstride); break;
        case 2: hipLaunchKernelGGL((norm_bwd_kernel<2, RMS>), grid, blk, 0, st, x, gamma, mean, rstd, dy, dx, dx_beta, pg, pb, rows, dx_in, rps, ngroups, pstride); break;
        case 3: hipLaunchKernelGGL((norm_bwd_kernel<3, RMS>), grid, blk, 0, st, x, gamma, mean, rstd, dy, dx, dx_beta, pg, pb, rows, dx_in, rps, ngroups, pstride); break;
        case 4: hipLaunchKernelGGL((norm_bwd_kernel<4, RMS>), grid, blk, 0, st, x, gamma, mean, rstd, dy, dx, dx_beta, pg, pb, rows, dx_in, rps, ngroups, pstride); break;
        case 8: hipLaunchKernelGGL((norm_bwd_kernel<8, RMS>), grid, blk, 0, st, x, gamma, mean, rstd, dy, dx, dx_beta, pg, pb, rows, dx_in, rps, ngroups, pstride); break;
        default: dyn::set_error("norm: unsupported C=%lld", (long long)C); return DYN_E_UNSUPPORTED;
    }
    // one reduction per sample into its group's gradient; a group's second sample accumulates onto the first (recording order = sample order)
    for (int sidx = 0; sidx < nsamp; ++sidx) {
        const int64_t po = (int64_t)(sidx % ngroups) * pstride;
        const float wb = sidx < ngroups ? wbeta : 1.f;
        float* pgs = pg + (int64_t)sidx * nbs * C;
        float* pbs = pb + (int64_t)sidx * nbs * C;
        if (!RMS && dgamma && dbeta) dyn::reduce_pair_or_defer(pgs, dgamma + po, pbs, dbeta + po, (int64_t)nbs, C, wb, st);
        else {
            if (dgamma) dyn::reduce_or_defer(pgs, dgamma + po, (int64_t)nbs, C, wb, st);
            if (!RMS && dbeta) dyn::reduce_or_defer(pbs, dbeta + po, (int64_t)nbs, C, wb, st);
        }
    }
    return dyn::check_launch("dyn_norm_bwd");
}

}  // namespace

// ---- per-channel affine with fixed statistics: BatchRenorm1d in eval mode (the dynamic-eval loop calls model.eval() so the
// running statistics are never updated, reference lcasr/lib.py:525):  y = (x - mean_c) * rsqrt(var_c + eps) * w_c + b_c
namespace {
__global__ __launch_bounds__(256) void chanaffine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ var, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ y, int64_t rows, int C,
                                                             float eps) {
    for (int c = threadIdx.x; c < C; c += 256) {
        const float m = mean[c], rs = rsqrtf(var[c] + eps), g = w[c], sh = b ? b[c] : 0.f;
        for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) y[r * C + c] = (x[r * C + c] - m) * rs * g + sh;
    }
}

// dx = dy * rs * w (+ dx_beta * dx); per-workgroup partial column sums of dy * xhat and dy, rows visited in a fixed order
__global__ __launch_bounds__(256) void chanaffine_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ var, const float* __restrict__ w,
                                                             const float* __restrict__ dy, float* __restrict__ dx, float dx_beta,
                                                             float* __restrict__ pg, float* __restrict__ pb, int64_t rows, int C,
                                                             float eps) {
    for (int c = threadIdx.x; c < C; c += 256) {
        const float m = mean[c], rs = rsqrtf(var[c] + eps), g = w[c];
        float ag = 0.f, ab = 0.f;
        for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
            const float gy = dy[r * C + c];
            ag += gy * ((x[r * C + c] - m) * rs);
            ab += gy;
            const float o = gy * rs * g;
            dx[r * C + c] = dx_beta != 0.f ? o + dx_beta * dx[r * C + c] : o;
        }
        pg[(int64_t)blockIdx.x * C + c] = ag;
        pb[(int64_t)blockIdx.x * C + c] = ab;
    }
}
inline int chan_blocks(int64_t rows) { return (int)(rows < 1 ? 1 : rows > 512 ? 512 : rows); }
}  // namespace

extern "C" int dyn_chanaffine_fwd(const float* x, const float* mean, const float* var, const float* weight, const float* bias,
                                  float* y, int64_t rows, int64_t C, float eps, void* stream) {
    DYN_REQUIRE(x && mean && var && weight && y && rows >= 0 && C > 0 && C < (1 << 30), DYN_E_ARG, "dyn_chanaffine_fwd: bad arguments");
    if (rows == 0) return DYN_OK;
    hipLaunchKernelGGL(chanaffine_fwd_kernel, dim3(chan_blocks(rows)), dim3(256), 0, (hipStream_t)stream, x, mean, var, weight, bias, y,
                       rows, (int)C, eps);
    return dyn::check_launch("dyn_chanaffine_fwd");
}

extern "C" int64_t dyn_chanaffine_bwd_workspace_bytes(int64_t rows, int64_t C) {
    return (int64_t)2 * chan_blocks(rows) * C * (int64_t)sizeof(float);
}

extern "C" int dyn_chanaffine_bwd(const float* x, const float* mean, const float* var, const float* weight, const float* dy,
                                  float* dx, float dx_beta, float* dweight, float* dbias, float wgrad_beta, int64_t rows, int64_t C,
                                  float eps, void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(x && mean && var && weight && dy && dx && rows >= 0 && C > 0 && C < (1 << 30), DYN_E_ARG, "dyn_chanaffine_bwd: bad arguments");
    if (rows == 0) return DYN_OK;
    const int nb = chan_blocks(rows);
    DYN_REQUIRE(workspace && workspace_bytes >= dyn_chanaffine_bwd_workspace_bytes(rows, C), DYN_E_WORKSPACE,
                "dyn_chanaffine_bwd: workspace too small");
    float* pg = dyn::partials_alloc(workspace, (int64_t)2 * nb * C * (int64_t)sizeof(float));
    float* pb = pg + (int64_t)nb * C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chanaffine_bwd_kernel, dim3(nb), dim3(256), 0, st, x, mean, var, weight, dy, dx, dx_beta, pg, pb, rows, (int)C, eps);
    if (dweight && dbias) dyn::reduce_pair_or_defer(pg, dweight, pb, dbias, (int64_t)nb, C, wgrad_beta, st);
    else {
        if (dweight) dyn::reduce_or_defer(pg, dweight, (int64_t)nb, C, wgrad_beta, st);
        if (dbias) dyn::reduce_or_defer(pb, dbias, (int64_t)nb, C, wgrad_beta, st);
    }
    return dyn::check_launch("dyn_chanaffine_bwd");
}

extern "C" int64_t dyn_norm_bwd_workspace_bytes(int64_t rows, int64_t C) {
    return (int64_t)2 * bwd_blocks(rows) * C * (int64_t)sizeof(float);
}

extern "C" int64_t dyn_norm_bwd_workspace_bytes_g(int64_t rows, int64_t C, int64_t rows_per_sample) {
    const int64_t rps = rows_per_sample > 0 ? rows_per_sample : rows;
    return (int64_t)2 * bwd_blocks(rps) * dyn::cdiv(rows, rps > 0 ? rps : 1) * C * (int64_t)sizeof(float);
}

// ---- lockstep-group variants: `rows` = n_samples * rows_per_sample rows; sample s uses the parameters of group s % n_groups, stored
// param_stride elements apart (the flat parameter buffers of the group's model replicas); weight gradients go to dgamma + group * param_stride.
extern "C" int dyn_layernorm_fwd_g(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                   int64_t rows, int64_t C, float eps, int64_t rows_per_sample, int64_t n_groups, int64_t param_stride,
                                   void* stream) {
    DYN_REQUIRE(x && gamma && y && mean && rstd && rows >= 0 && C > 0 && C % 256 == 0 && rows_per_sample > 0 && n_groups >= 1 &&
                    rows % rows_per_sample == 0 && (rows / rows_per_sample) % n_groups == 0 && param_stride % 4 == 0,
                DYN_E_ARG, "dyn_layernorm_fwd_g: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_fwd<false>(x, gamma, beta, y, mean, rstd, rows, C, eps, (hipStream_t)stream, Grouping{rows_per_sample, (int)n_groups, param_stride});
}

extern "C" int dyn_layernorm_bwd_g(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy,
                                   const float* dx_in, float* dx, float dx_beta, float* dgamma, float* dbeta, float wgrad_beta,
                                   int64_t rows, int64_t C, int64_t rows_per_sample, int64_t n_groups, int64_t param_stride,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(x && gamma && mean && rstd && dy && dx && rows >= 0 && C > 0 && C % 256 == 0 && rows_per_sample > 0 && n_groups >= 1 &&
                    rows % rows_per_sample == 0 && (rows / rows_per_sample) % n_groups == 0 && param_stride % 4 == 0,
                DYN_E_ARG, "dyn_layernorm_bwd_g: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_bwd<false>(x, gamma, mean, rstd, dy, dx, dx_beta, dgamma, dbeta, wgrad_beta, rows, C, workspace, workspace_bytes,
                             (hipStream_t)stream, dx_in, Grouping{rows_per_sample, (int)n_groups, param_stride});
}

extern "C" int dyn_rmsnorm_bwd_g(const float* x, const float* gamma, const float* rstd, const float* dy, float* dx, float dx_beta,
                                 float* dgamma, float wgrad_beta, int64_t rows, int64_t C, int64_t rows_per_sample, int64_t n_groups,
                                 int64_t param_stride, void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(x && gamma && rstd && dy && dx && rows >= 0 && C > 0 && C % 256 == 0 && rows_per_sample > 0 && n_groups >= 1 &&
                    rows % rows_per_sample == 0 && (rows / rows_per_sample) % n_groups == 0 && param_stride % 4 == 0,
                DYN_E_ARG, "dyn_rmsnorm_bwd_g: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_bwd<true>(x, gamma, nullptr, rstd, dy, dx, dx_beta, dgamma, nullptr, wgrad_beta, rows, C, workspace, workspace_bytes,
                            (hipStream_t)stream, nullptr, Grouping{rows_per_sample, (int)n_groups, param_stride});
}

extern "C" int dyn_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                 int64_t rows, int64_t C, float eps, void* stream) {
    DYN_REQUIRE(x && gamma && y && mean && rstd && rows >= 0 && C > 0 && C % 256 == 0, DYN_E_ARG,
                "dyn_layernorm_fwd: bad arguments (C=%lld must be a multiple of 256)", (long long)C);
    if (rows == 0) return DYN_OK;
    return launch_fwd<false>(x, gamma, beta, y, mean, rstd, rows, C, eps, (hipStream_t)stream);
}

extern "C" int dyn_layernorm_bwd(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy,
                                 float* dx, float dx_beta, float* dgamma, float* dbeta, float wgrad_beta, int64_t rows,
                                 int64_t C, void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(x && gamma && mean && rstd && dy && dx && rows >= 0 && C > 0 && C % 256 == 0, DYN_E_ARG,
                "dyn_layernorm_bwd: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_bwd<false>(x, gamma, mean, rstd, dy, dx, dx_beta, dgamma, dbeta, wgrad_beta, rows, C, workspace,
                             workspace_bytes, (hipStream_t)stream);
}

extern "C" int dyn_layernorm_bwd_res(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy,
                                     const float* dx_in, float* dx, float dx_beta, float* dgamma, float* dbeta, float wgrad_beta,
                                     int64_t rows, int64_t C, void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(x && gamma && mean && rstd && dy && dx && dx_in && rows >= 0 && C > 0 && C % 256 == 0, DYN_E_ARG,
                "dyn_layernorm_bwd_res: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_bwd<false>(x, gamma, mean, rstd, dy, dx, dx_beta, dgamma, dbeta, wgrad_beta, rows, C, workspace,
                             workspace_bytes, (hipStream_t)stream, dx_in);
}

extern "C" int dyn_rmsnorm_fwd(const float* x, const float* gamma, float* y, float* rstd, int64_t rows, int64_t C, float eps,
                               void* stream) {
    DYN_REQUIRE(x && gamma && y && rstd && rows >= 0 && C > 0 && C % 256 == 0, DYN_E_ARG, "dyn_rmsnorm_fwd: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_fwd<true>(x, gamma, nullptr, y, nullptr, rstd, rows, C, eps, (hipStream_t)stream);
}

extern "C" int dyn_rmsnorm_bwd(const float* x, const float* gamma, const float* rstd, const float* dy, float* dx,
                               float dx_beta, float* dgamma, float wgrad_beta, int64_t rows, int64_t C, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(x && gamma && rstd && dy && dx && rows >= 0 && C > 0 && C % 256 == 0, DYN_E_ARG, "dyn_rmsnorm_bwd: bad arguments");
    if (rows == 0) return DYN_OK;
    return launch_bwd<true>(x, gamma, nullptr, rstd, dy, dx, dx_beta, dgamma, nullptr, wgrad_beta, rows, C, workspace,
                            workspace_bytes, (hipStream_t)stream);
}
