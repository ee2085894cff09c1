// Convolutions of the conformer encoder, channels-last so every access is coalesced along C:
//   * depthwise Conv1d (kernel KW <= 31, 'same' zero padding) of the conformer conv module, fwd / dgrad / wgrad;
//   * the dw_striding x8 subsampling front-end: first 3x3 stride-2 conv (1 -> C channels, direct) and the
//     3x3 stride-2 depthwise convs with the preceding SiLU fused into the load (pointwise 1x1 convs are GEMMs).
// These are HBM-bound (a few FLOPs per byte): each thread owns one channel and slides over time so every
// input element is loaded once per workgroup and reused from registers.
// Reference call sites: upstream SCConformerXL conv module / subsampling reached through
// model(audio_signal=...) (reference lcasr/lib.py:164,550; earnings_finetune/lcasr160rb1.yaml:10-15).
#include "common.h"
#include "reduce.h"

namespace {

__device__ __forceinline__ float silu_f(float x) { return x * dyn::sigmoidf_(x); }
__device__ __forceinline__ float silu_grad(float x) {
    const float s = dyn::sigmoidf_(x);
    return s * (1.f + x * (1.f - s));
}

constexpr int TT = 32;  // time steps per workgroup tile

// y[b,t,c] = bias[c] + sum_j w[c,j] * x[b, t + j - P, c],  P = (KW-1)/2.   FLIP => dgrad (w index reversed).
template <int KW, bool FLIP>
__global__ __launch_bounds__(256) void dwconv1d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int64_t T,
                                                        int C, float beta, int ngroups, int64_t pstride) {
    constexpr int P = (KW - 1) / 2;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t b = blockIdx.z;
    {   // lockstep group: sample b convolves with the filters of replica b % ngroups (ngroups = 1: plain launch)
        const int64_t poff = (int64_t)(blockIdx.z % ngroups) * pstride;
        w += poff;
        if (bias) bias += poff;
    }
    const int64_t t0 = (int64_t)blockIdx.x * TT;
    float wk[KW];
#pragma unroll
    for (int j = 0; j < KW; ++j) wk[j] = w[(int64_t)c * KW + (FLIP ? KW - 1 - j : j)];
    const float bv = bias ? bias[c] : 0.f;
    const float* xb = x + b * T * C + c;
    float* yb = y + b * T * C + c;
    float win[KW];
#pragma unroll
    for (int j = 0; j < KW - 1; ++j) {
        const int64_t t = t0 + j - P;
        win[j + 1] = (t >= 0 && t < T) ? xb[t * C] : 0.f;
    }
    for (int k = 0; k < TT; ++k) {
        const int64_t t = t0 + k;
        if (t >= T) break;
#pragma unroll
        for (int j = 0; j < KW - 1; ++j) win[j] = win[j + 1];
        const int64_t tn = t + P;
        win[KW - 1] = tn < T ? xb[tn * C] : 0.f;
        float acc = bv;
#pragma unroll
        for (int j = 0; j < KW; ++j) acc += wk[j] * win[j];
        yb[t * C] = beta != 0.f ? acc + beta * yb[t * C] : acc;
    }
}

// partial[(tile), j, c] = sum_{t in tile} dy[b,t,c] * x[b, t + j - P, c];   partial_b[(tile), c] = sum dy
template <int KW>
__global__ __launch_bounds__(256) void dwconv1d_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ partial_w, float* __restrict__ partial_b,
                                                              int64_t T, int C, int64_t t_per_block) {
    constexpr int P = (KW - 1) / 2;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t b = blockIdx.z;
    const int64_t t0 = (int64_t)blockIdx.x * t_per_block;
    const int64_t t1 = (t0 + t_per_block < T) ? t0 + t_per_block : T;
    const float* xb = x + b * T * C + c;
    const float* gb = dy + b * T * C + c;
    float acc[KW], win[KW];
    float accb = 0.f;
#pragma unroll
    for (int j = 0; j < KW; ++j) acc[j] = 0.f;
#pragma unroll
    for (int j = 0; j < KW - 1; ++j) {
        const int64_t t = t0 + j - P;
        win[j + 1] = (t >= 0 && t < T) ? xb[t * C] : 0.f;
    }
    for (int64_t t = t0; t < t1; ++t) {
#pragma unroll
        for (int j = 0; j < KW - 1; ++j) win[j] = win[j + 1];
        const int64_t tn = t + P;
        win[KW - 1] = tn < T ? xb[tn * C] : 0.f;
        const float g = gb[t * C];
        accb += g;
#pragma unroll
        for (int j = 0; j < KW; ++j) acc[j] += g * win[j];
    }
    const int64_t tile = (int64_t)blockIdx.z * gridDim.x + blockIdx.x;
#pragma unroll
    for (int j = 0; j < KW; ++j) partial_w[(tile * C + c) * KW + j] = acc[j];
    partial_b[tile * C + c] = accb;
}

// ---- subsampling: first conv, 1 input channel, 3x3, stride 2, pad 1:  x [B, T, F] -> z [B, To, Fo, C] ----
__global__ __launch_bounds__(256) void conv2d_first_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ z,
                                                                int64_t T, int F, int64_t To, int Fo, int C) {
    // One workgroup = one (b, to) row, looping over fo; thread = channel (C <= 256 per x-block).
    const int c = blockIdx.y * 256 + threadIdx.x;
    const int64_t to = blockIdx.x, b = blockIdx.z;
    if (c >= C) return;
    float wk[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) wk[j] = w[c * 9 + j];
    const float bv = bias[c];
    const float* xb = x + b * T * F;
    float* zb = z + ((b * To + to) * Fo) * C + c;
    for (int fo = 0; fo < Fo; ++fo) {
        float acc = bv;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int64_t t = 2 * to + dt - 1;
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                const int f = 2 * fo + df - 1;
                const float v = (t >= 0 && t < T && f >= 0 && f < F) ? xb[t * F + f] : 0.f;  // wave-uniform load
                acc += wk[dt * 3 + df] * v;
            }
        }
        zb[(int64_t)fo * C] = acc;
    }
}

// dgrad of the first conv (only the entropy-gradient input perturbation needs it, reference lcasr/lib.py:86-99):
//   dx[b,t,f] = sum_c sum_{(to,dt),(fo,df): 2to+dt-1 = t, 2fo+df-1 = f} w[c,dt,df] * dz[b,to,fo,c]
// one wave per output element: lanes stride the channels, wavefront reduction at the end.
__global__ __launch_bounds__(256) void conv2d_first_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                                  float* __restrict__ dx, int64_t B, int64_t T, int F,
                                                                  int64_t To, int Fo, int C) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t total = B * T * F;
    for (int64_t e = (int64_t)blockIdx.x * 4 + wv; e < total; e += (int64_t)gridDim.x * 4) {
        const int f = (int)(e % F);
        const int64_t t = (e / F) % T, b = e / ((int64_t)F * T);
        float acc = 0.f;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int64_t tt = t + 1 - dt;
            if (tt < 0 || (tt & 1) || (tt >> 1) >= To) continue;
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                const int ff = f + 1 - df;
                if (ff < 0 || (ff & 1) || (ff >> 1) >= Fo) continue;
                const float* g = dz + ((b * To + (tt >> 1)) * Fo + (ff >> 1)) * C;
                for (int c = lane; c < C; c += 64) acc += w[c * 9 + dt * 3 + df] * g[c];
            }
        }
        acc = dyn::wave_sum(acc);
        if (lane == 0) dx[e] = acc;
    }
}

// wgrad of the first conv: partial[(b,to-chunk), c, 9] and bias partial.
__global__ __launch_bounds__(256) void conv2d_first_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                  float* __restrict__ partial_w, float* __restrict__ partial_b,
                                                                  int64_t T, int F, int64_t To, int Fo, int C,
                                                                  int64_t to_per_block) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    const int64_t b = blockIdx.z;
    if (c >= C) return;
    const int64_t to0 = (int64_t)blockIdx.x * to_per_block;
    const int64_t to1 = (to0 + to_per_block < To) ? to0 + to_per_block : To;
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0.f;
    float accb = 0.f;
    const float* xb = x + b * T * F;
    for (int64_t to = to0; to < to1; ++to) {
        const float* gz = dz + ((b * To + to) * Fo) * C + c;
        for (int fo = 0; fo < Fo; ++fo) {
            const float g = gz[(int64_t)fo * C];
            accb += g;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int64_t t = 2 * to + dt - 1;
#pragma unroll
                for (int df = 0; df < 3; ++df) {
                    const int f = 2 * fo + df - 1;
                    const float v = (t >= 0 && t < T && f >= 0 && f < F) ? xb[t * F + f] : 0.f;
                    acc[dt * 3 + df] += g * v;
                }
            }
        }
    }
    const int64_t tile = (int64_t)blockIdx.z * gridDim.x + blockIdx.x;
#pragma unroll
    for (int j = 0; j < 9; ++j) partial_w[(tile * C + c) * 9 + j] = acc[j];
    partial_b[tile * C + c] = accb;
}

// ---- subsampling: depthwise 3x3 stride 2 pad 1 over (T, F), input activation SiLU fused into the load ----
//   u[b,to,fo,c] = bias[c] + sum_{dt,df} w[c,dt,df] * silu(z[b, 2to+dt-1, 2fo+df-1, c])
__global__ __launch_bounds__(256) void dwconv2d_s2_fwd_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ u,
                                                               int64_t T, int F, int64_t To, int Fo, int C) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    const int64_t to = blockIdx.x, b = blockIdx.z;
    if (c >= C) return;
    float wk[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) wk[j] = w[c * 9 + j];
    const float bv = bias[c];
    const float* zb = z + b * T * F * C + c;
    float* ub = u + ((b * To + to) * Fo) * C + c;
    for (int fo = 0; fo < Fo; ++fo) {
        float acc = bv;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int64_t t = 2 * to + dt - 1;
            if (t < 0 || t >= T) continue;
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                const int f = 2 * fo + df - 1;
                if (f < 0 || f >= F) continue;
                acc += wk[dt * 3 + df] * silu_f(zb[(t * F + f) * C]);
            }
        }
        ub[(int64_t)fo * C] = acc;
    }
}


// ---- vectorised subsampling forwards (C % 4 == 0): one wave per output row (b, to), lane = 4 consecutive channels, so every
// access is a 16-B-per-lane transaction (1 KiB per wave-instruction), and the 3-column window slides along f: each input
// element is loaded (and SiLU'd) once per output row instead of up to 3 times.  Accumulation order per output is the same
// (dt major, df minor) as the scalar kernels, so results are bit-identical to them.
__device__ __forceinline__ float4 silu4(float4 v) { return make_float4(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w)); }
__device__ __forceinline__ void fma4(float4& a, const float4& w, const float4& v) {
    a.x += w.x * v.x; a.y += w.y * v.y; a.z += w.z * v.z; a.w += w.w * v.w;
}
__device__ __forceinline__ void load_w4(const float* __restrict__ w, int c0, float4 (&wk)[9]) {
#pragma unroll
    for (int j = 0; j < 9; ++j) wk[j] = make_float4(w[(c0 + 0) * 9 + j], w[(c0 + 1) * 9 + j], w[(c0 + 2) * 9 + j], w[(c0 + 3) * 9 + j]);
}

__global__ __launch_bounds__(256) void dwconv2d_s2_fwd_v4_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ u,
                                                                  int64_t T, int F, int64_t To, int Fo, int C, int64_t rows) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;     // (b, to) flattened
    if (row >= rows) return;
    const int64_t b = row / To, to = row % To;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
        float4 wk[9];
        load_w4(w, c0, wk);
        const float4 bv = *reinterpret_cast<const float4*>(bias + c0);
        const float* zr[3];
        bool ok[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int64_t t = 2 * to + dt - 1;
            ok[dt] = t >= 0 && t < T;
            zr[dt] = z + ((b * T + (ok[dt] ? t : 0)) * F) * C + c0;
        }
        float4 prev[3] = {zero, zero, zero};      // column 2*fo - 1 (f = -1 for fo = 0: padding)
        float* ur = u + (row * Fo) * C + c0;
        for (int fo = 0; fo < Fo; ++fo) {
            const int f0 = 2 * fo, f1 = 2 * fo + 1;
            float4 c0v[3], c1v[3];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                c0v[dt] = ok[dt] ? silu4(*reinterpret_cast<const float4*>(zr[dt] + (int64_t)f0 * C)) : zero;
                c1v[dt] = (ok[dt] && f1 < F) ? silu4(*reinterpret_cast<const float4*>(zr[dt] + (int64_t)f1 * C)) : zero;
            }
            float4 acc = bv;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                // skipped taps of the scalar kernel are exact zeros here: adding w * 0 leaves the sum unchanged
                fma4(acc, wk[dt * 3 + 0], prev[dt]);
                fma4(acc, wk[dt * 3 + 1], c0v[dt]);
                fma4(acc, wk[dt * 3 + 2], c1v[dt]);
                prev[dt] = c1v[dt];
            }
            *reinterpret_cast<float4*>(ur + (int64_t)fo * C) = acc;
        }
    }
}

__global__ __launch_bounds__(256) void conv2d_first_fwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, float* __restrict__ z,
                                                                   int64_t T, int F, int64_t To, int Fo, int C, int64_t rows) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;
    if (row >= rows) return;
    const int64_t b = row / To, to = row % To;
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
        float4 wk[9];
        load_w4(w, c0, wk);
        const float4 bv = *reinterpret_cast<const float4*>(bias + c0);
        const float* xr[3];
        bool ok[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int64_t t = 2 * to + dt - 1;
            ok[dt] = t >= 0 && t < T;
            xr[dt] = x + (b * T + (ok[dt] ? t : 0)) * F;
        }
        float prev[3] = {0.f, 0.f, 0.f};
        float* zr = z + (row * Fo) * C + c0;
        for (int fo = 0; fo < Fo; ++fo) {
            const int f0 = 2 * fo, f1 = 2 * fo + 1;
            float4 acc = bv;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const float a0 = ok[dt] ? xr[dt][f0] : 0.f;                 // wave-uniform loads
                const float a1 = (ok[dt] && f1 < F) ? xr[dt][f1] : 0.f;
                const float p = prev[dt];
                acc.x += wk[dt * 3 + 0].x * p; acc.y += wk[dt * 3 + 0].y * p; acc.z += wk[dt * 3 + 0].z * p; acc.w += wk[dt * 3 + 0].w * p;
                acc.x += wk[dt * 3 + 1].x * a0; acc.y += wk[dt * 3 + 1].y * a0; acc.z += wk[dt * 3 + 1].z * a0; acc.w += wk[dt * 3 + 1].w * a0;
                acc.x += wk[dt * 3 + 2].x * a1; acc.y += wk[dt * 3 + 2].y * a1; acc.z += wk[dt * 3 + 2].z * a1; acc.w += wk[dt * 3 + 2].w * a1;
                prev[dt] = a1;
            }
            *reinterpret_cast<float4*>(zr + (int64_t)fo * C) = acc;
        }
    }
}


// ---- vectorised subsampling backwards (C % 4 == 0): same wave-per-row / lane = 4 channels scheme as the forwards ----
__device__ __forceinline__ float4 silu_grad4(float4 v) { return make_float4(silu_grad(v.x), silu_grad(v.y), silu_grad(v.z), silu_grad(v.w)); }

// dz row (b, t): every input element gathers the 1, 2 or 4 outputs it fed (parity of t and f decides which taps exist).
__global__ __launch_bounds__(256) void dwconv2d_s2_dgrad_v4_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                                    const float* __restrict__ du, float* __restrict__ dz,
                                                                    int64_t T, int F, int64_t To, int Fo, int C, int64_t rows) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;     // (b, t) flattened
    if (row >= rows) return;
    const int64_t b = row / T, t = row % T;
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
        float4 wk[9];
        load_w4(w, c0, wk);
        const float* zr = z + (row * F) * C + c0;
        float* dr = dz + (row * F) * C + c0;
        const float* gb = du + (b * To * Fo) * C + c0;
        for (int f = 0; f < F; ++f) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int64_t tt = t + 1 - dt;  // = 2 * to
                if (tt < 0 || (tt & 1)) continue;
                const int64_t to = tt >> 1;
                if (to >= To) continue;
#pragma unroll
                for (int df = 0; df < 3; ++df) {
                    const int ff = f + 1 - df;
                    if (ff < 0 || (ff & 1)) continue;
                    const int fo = ff >> 1;
                    if (fo >= Fo) continue;
                    fma4(acc, wk[dt * 3 + df], *reinterpret_cast<const float4*>(gb + (to * Fo + fo) * C));
                }
            }
            const float4 sg = silu_grad4(*reinterpret_cast<const float4*>(zr + (int64_t)f * C));
            *reinterpret_cast<float4*>(dr + (int64_t)f * C) = make_float4(acc.x * sg.x, acc.y * sg.y, acc.z * sg.z, acc.w * sg.w);
        }
    }
}

// weight gradient partials: one wave per tile of `per` output rows, sliding 3-column window like the forward; the four
// waves of a workgroup are summed in wave order through LDS, so one partial row per WORKGROUP reaches the reducer.
template <bool FIRST>   // FIRST: the 1-channel first conv (input x [B, T, F], no activation); else depthwise with SiLU(z)
__global__ __launch_bounds__(256) void conv2d_s2_wgrad_v4_kernel(const float* __restrict__ in, const float* __restrict__ g_out,
                                                                  float* __restrict__ partial_w, float* __restrict__ partial_b,
                                                                  int64_t T, int F, int64_t To, int Fo, int C, int64_t per,
                                                                  int64_t chunks, int64_t tiles) {
    __shared__ float4 red[3][64][10];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;     // (b, chunk) flattened
    const bool live = tile < tiles;
    const int64_t b = live ? tile / chunks : 0, to0 = live ? (tile % chunks) * per : 0;
    const int64_t to1 = live ? ((to0 + per < To) ? to0 + per : To) : 0;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int cb = 0; cb < C; cb += 256) {
        const int c0 = cb + lane * 4;
        const bool cok = c0 < C;
        float4 acc[9], accb = zero;
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] = zero;
        for (int64_t to = to0; to < to1 && cok; ++to) {
            const float* ir[3];
            bool ok[3];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int64_t t = 2 * to + dt - 1;
                ok[dt] = t >= 0 && t < T;
                ir[dt] = FIRST ? in + (b * T + (ok[dt] ? t : 0)) * F : in + ((b * T + (ok[dt] ? t : 0)) * F) * C + c0;
            }
            const float* gr = g_out + ((b * To + to) * Fo) * C + c0;
            float4 prev[3] = {zero, zero, zero};
            for (int fo = 0; fo < Fo; ++fo) {
                const int f0 = 2 * fo, f1 = 2 * fo + 1;
                const float4 g = *reinterpret_cast<const float4*>(gr + (int64_t)fo * C);
                accb.x += g.x; accb.y += g.y; accb.z += g.z; accb.w += g.w;
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    float4 a0 = zero, a1 = zero;
                    if (FIRST) {
                        if (ok[dt]) { const float v = ir[dt][f0]; a0 = make_float4(v, v, v, v); }
                        if (ok[dt] && f1 < F) { const float v = ir[dt][f1]; a1 = make_float4(v, v, v, v); }
                    } else {
                        if (ok[dt]) a0 = silu4(*reinterpret_cast<const float4*>(ir[dt] + (int64_t)f0 * C));
                        if (ok[dt] && f1 < F) a1 = silu4(*reinterpret_cast<const float4*>(ir[dt] + (int64_t)f1 * C));
                    }
                    fma4(acc[dt * 3 + 0], g, prev[dt]);
                    fma4(acc[dt * 3 + 1], g, a0);
                    fma4(acc[dt * 3 + 2], g, a1);
                    prev[dt] = a1;
                }
            }
        }
        __syncthreads();
        if (wv > 0) {
#pragma unroll
            for (int j = 0; j < 9; ++j) red[wv - 1][lane][j] = acc[j];
            red[wv - 1][lane][9] = accb;
        }
        __syncthreads();
        if (wv == 0 && cok) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int j = 0; j < 9; ++j) { const float4 v = red[k][lane][j]; acc[j].x += v.x; acc[j].y += v.y; acc[j].z += v.z; acc[j].w += v.w; }
                const float4 v = red[k][lane][9];
                accb.x += v.x; accb.y += v.y; accb.z += v.z; accb.w += v.w;
            }
            // partial rows are tap-major ([workgroup][j][C]): every store is a coalesced 16 B per lane; the reducer
            // (reduce_partials_2d_taps_kernel) writes the [C][3][3] weight-gradient layout
            const int64_t prow = blockIdx.x;
#pragma unroll
            for (int j = 0; j < 9; ++j) *reinterpret_cast<float4*>(partial_w + (prow * 9 + j) * C + c0) = acc[j];
            *reinterpret_cast<float4*>(partial_b + prow * C + c0) = accb;
        }
    }
}

// ---- the first two subsampling stages FUSED: u2 = dw3x3_s2(silu(conv3x3_s2(x))) without the [B, T/2, F/2, C] intermediate ----
// z1 = conv2d_first(x) is the largest activation of the model (671 MB at B = 2, T = 16384, C = 256) and is a pure function of the
// one-channel input x (5 MB): writing it, reading it back for the depthwise conv and reading it twice more in the backward (plus the
// 335 MB of its gradient, written and read once) is 2.7 GB of HBM traffic per adapt step for ~10 GFLOP.  The fused kernels
// RECOMPUTE z1 from x on the fly (9 FMAs per element, x arrives through the scalar cache: the row index is wave-uniform) — forward
// and backward — so neither z1 nor dz1 ever exists in HBM.  Same accumulation orders as the unfused kernels above (bias first, taps
// dt-major): the forward output is bit-identical to conv2d_first_fwd_v4 + dwconv2d_s2_fwd_v4.
//   forward : one wave per output row (b, t2), lane = 4 channels, sliding 3-column window of silu(z1) along f
//   backward: one wave per tile of z1 rows; per z1 element: z1 again, dz1 = silu'(z1) * sum w2 * du2 (never stored), and straight
//             into the four weight / bias gradient accumulators (conv1: dz1 * x taps; dw2: du2 * silu(z1)); partial rows per
//             workgroup, summed in a fixed order by the 2-D reducers (deterministic, no atomics)
__device__ __forceinline__ void fma4s(float4& a, const float4& w, float v) {
    a.x += w.x * v; a.y += w.y * v; a.z += w.z * v; a.w += w.w * v;
}

__global__ __launch_bounds__(256) void sub12_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ u2, int64_t T, int F,
                                                         int64_t T1, int F1, int64_t T2, int F2, int C, int64_t rows) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wave-uniform: everything derived from `row` stays scalar
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;                       // (b, t2) flattened
    if (row >= rows) return;
    const int64_t b = row / T2, t2 = row % T2;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    // x rows 4 t2 - 3 .. 4 t2 + 3 feed the three z1 rows 2 t2 - 1 .. 2 t2 + 1
    const float* xr[7];
    bool xok[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const int64_t tx = 4 * t2 - 3 + r;
        xok[r] = tx >= 0 && tx < T;
        xr[r] = x + (b * T + (xok[r] ? tx : 0)) * F;
    }
    bool rok[3];
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
        const int64_t t1 = 2 * t2 + dt - 1;
        rok[dt] = t1 >= 0 && t1 < T1;
    }
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
        float4 wa[9], wb[9];
        load_w4(w1, c0, wa);
        load_w4(w2, c0, wb);
        const float4 b1v = *reinterpret_cast<const float4*>(b1 + c0);
        const float4 b2v = *reinterpret_cast<const float4*>(b2 + c0);
        float4 prev[3] = {zero, zero, zero};                  // silu(z1) at column 2 f2 - 1 (f1 = -1 for f2 = 0: padding)
        float xprev[7];                                        // x column 4 f2 - 1
#pragma unroll
        for (int r = 0; r < 7; ++r) xprev[r] = 0.f;
        float* ur = u2 + (row * F2) * C + c0;
        for (int f2 = 0; f2 < F2; ++f2) {
            float xv[7][5];                                    // x columns 4 f2 - 1 .. 4 f2 + 3
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                xv[r][0] = xprev[r];
#pragma unroll
                for (int k = 1; k < 5; ++k) {
                    const int col = 4 * f2 - 1 + k;
                    xv[r][k] = (xok[r] && col < F) ? xr[r][col] : 0.f;
                }
                xprev[r] = xv[r][4];
            }
            float4 s[3][2];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float4 z = b1v;
#pragma unroll
                    for (int dtx = 0; dtx < 3; ++dtx)
#pragma unroll
                        for (int dfx = 0; dfx < 3; ++dfx) fma4s(z, wa[dtx * 3 + dfx], xv[2 * dt + dtx][2 * j + dfx]);
                    s[dt][j] = (rok[dt] && 2 * f2 + j < F1) ? silu4(z) : zero;
                }
            float4 acc = b2v;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                fma4(acc, wb[dt * 3 + 0], prev[dt]);
                fma4(acc, wb[dt * 3 + 1], s[dt][0]);
                fma4(acc, wb[dt * 3 + 2], s[dt][1]);
                prev[dt] = s[dt][1];
            }
            *reinterpret_cast<float4*>(ur + (int64_t)f2 * C) = acc;
        }
    }
}

__global__ __launch_bounds__(256) void sub12_bwd_kernel(const float* __restrict__ x, const float* __restrict__ du2,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, float* __restrict__ pw1, float* __restrict__ pb1,
                                                         float* __restrict__ pw2, float* __restrict__ pb2, int64_t T, int F, int64_t T1,
                                                         int F1, int64_t T2, int F2, int C, int64_t per, int64_t chunks, int64_t tiles) {
    __shared__ float4 red[3][64][10];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t tile = (int64_t)blockIdx.x * 4 + wv;     // (b, chunk of z1 rows) flattened
    const bool live = tile < tiles;
    const int64_t b = live ? tile / chunks : 0, r0 = live ? (tile % chunks) * per : 0;
    const int64_t r1 = live ? ((r0 + per < T1) ? r0 + per : T1) : 0;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int cb = 0; cb < C; cb += 256) {
        const int c0 = cb + lane * 4;
        const bool cok = c0 < C;
        float4 wa[9], wb[9], a1[9], a2[9], ab1 = zero, ab2 = zero, b1v = zero;
#pragma unroll
        for (int j = 0; j < 9; ++j) { a1[j] = zero; a2[j] = zero; wa[j] = zero; wb[j] = zero; }
        if (cok) {
            load_w4(w1, c0, wa);
            load_w4(w2, c0, wb);
            b1v = *reinterpret_cast<const float4*>(b1 + c0);
        }
        for (int64_t t1 = r0; t1 < r1 && cok; ++t1) {
            const float* xr[3];
            bool xok[3];
#pragma unroll
            for (int dtx = 0; dtx < 3; ++dtx) {
                const int64_t tx = 2 * t1 + dtx - 1;
                xok[dtx] = tx >= 0 && tx < T;
                xr[dtx] = x + (b * T + (xok[dtx] ? tx : 0)) * F;
            }
            const float* gr[3];                       // du2 rows that read z1 row t1 through tap row dt:  2 t2 + dt - 1 = t1
            bool gok[3];
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int64_t tt = t1 + 1 - dt;
                gok[dt] = tt >= 0 && !(tt & 1) && (tt >> 1) < T2;
                gr[dt] = du2 + ((b * T2 + (gok[dt] ? (tt >> 1) : 0)) * F2) * C + c0;
            }
            for (int f1 = 0; f1 < F1; ++f1) {
                float xv[3][3];
#pragma unroll
                for (int dtx = 0; dtx < 3; ++dtx)
#pragma unroll
                    for (int dfx = 0; dfx < 3; ++dfx) {
                        const int col = 2 * f1 + dfx - 1;
                        xv[dtx][dfx] = (xok[dtx] && col >= 0 && col < F) ? xr[dtx][col] : 0.f;
                    }
                float4 z = b1v;
#pragma unroll
                for (int dtx = 0; dtx < 3; ++dtx)
#pragma unroll
                    for (int dfx = 0; dfx < 3; ++dfx) fma4s(z, wa[dtx * 3 + dfx], xv[dtx][dfx]);
                float4 sz, sg;                      // silu(z) and silu'(z) from ONE sigmoid per element (same values as silu_f / silu_grad)
                {
                    const float zz[4] = {z.x, z.y, z.z, z.w};
                    float a[4], d[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float sgm = dyn::sigmoidf_(zz[q]);
                        a[q] = zz[q] * sgm;
                        d[q] = sgm * (1.f + zz[q] * (1.f - sgm));
                    }
                    sz = make_float4(a[0], a[1], a[2], a[3]);
                    sg = make_float4(d[0], d[1], d[2], d[3]);
                }
                float4 g = zero;
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    if (!gok[dt]) continue;
#pragma unroll
                    for (int df = 0; df < 3; ++df) {
                        const int ff = f1 + 1 - df;
                        if (ff < 0 || (ff & 1) || (ff >> 1) >= F2) continue;
                        const float4 d = *reinterpret_cast<const float4*>(gr[dt] + (int64_t)(ff >> 1) * C);
                        fma4(g, wb[dt * 3 + df], d);
                        fma4(a2[dt * 3 + df], d, sz);
                        if (dt == 1 && df == 1) { ab2.x += d.x; ab2.y += d.y; ab2.z += d.z; ab2.w += d.w; }   // every du2 element once
                    }
                }
                const float4 dz = make_float4(g.x * sg.x, g.y * sg.y, g.z * sg.z, g.w * sg.w);
                ab1.x += dz.x; ab1.y += dz.y; ab1.z += dz.z; ab1.w += dz.w;
#pragma unroll
                for (int dtx = 0; dtx < 3; ++dtx)
#pragma unroll
                    for (int dfx = 0; dfx < 3; ++dfx) fma4s(a1[dtx * 3 + dfx], dz, xv[dtx][dfx]);
            }
        }
        // the four waves of the workgroup are summed in wave order through LDS (conv1 accumulators, then the depthwise ones)
        const int64_t prow = blockIdx.x;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float4* acc = pass == 0 ? a1 : a2;
            float4& accb = pass == 0 ? ab1 : ab2;
            __syncthreads();
            if (wv > 0) {
#pragma unroll
                for (int j = 0; j < 9; ++j) red[wv - 1][lane][j] = acc[j];
                red[wv - 1][lane][9] = accb;
            }
            __syncthreads();
            if (wv == 0 && cok) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
#pragma unroll
                    for (int j = 0; j < 9; ++j) { const float4 v = red[k][lane][j]; acc[j].x += v.x; acc[j].y += v.y; acc[j].z += v.z; acc[j].w += v.w; }
                    const float4 v = red[k][lane][9];
                    accb.x += v.x; accb.y += v.y; accb.z += v.z; accb.w += v.w;
                }
                float* pw = pass == 0 ? pw1 : pw2;
                float* pb = pass == 0 ? pb1 : pb2;
#pragma unroll
                for (int j = 0; j < 9; ++j) *reinterpret_cast<float4*>(pw + (prow * 9 + j) * C + c0) = acc[j];
                *reinterpret_cast<float4*>(pb + prow * C + c0) = accb;
            }
        }
    }
}

// dz[b,t,f,c] = silu'(z) * sum_{(to,dt),(fo,df): 2to+dt-1=t, 2fo+df-1=f} w[c,dt,df] * du[b,to,fo,c]
__global__ __launch_bounds__(256) void dwconv2d_s2_dgrad_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                                 const float* __restrict__ du, float* __restrict__ dz,
                                                                 int64_t T, int F, int64_t To, int Fo, int C) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    const int64_t t = blockIdx.x, b = blockIdx.z;
    if (c >= C) return;
    float wk[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) wk[j] = w[c * 9 + j];
    const float* gb = du + b * To * Fo * C + c;
    for (int f = 0; f < F; ++f) {
        float acc = 0.f;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int64_t tt = t + 1 - dt;  // = 2*to
            if (tt < 0 || (tt & 1)) continue;
            const int64_t to = tt >> 1;
            if (to >= To) continue;
#pragma unroll
            for (int df = 0; df < 3; ++df) {
                const int ff = f + 1 - df;
                if (ff < 0 || (ff & 1)) continue;
                const int fo = ff >> 1;
                if (fo >= Fo) continue;
                acc += wk[dt * 3 + df] * gb[(to * Fo + fo) * C];
            }
        }
        const int64_t idx = ((b * T + t) * F + f) * C + c;
        dz[idx] = acc * silu_grad(z[idx]);
    }
}

__global__ __launch_bounds__(256) void dwconv2d_s2_wgrad_kernel(const float* __restrict__ z, const float* __restrict__ du,
                                                                 float* __restrict__ partial_w, float* __restrict__ partial_b,
                                                                 int64_t T, int F, int64_t To, int Fo, int C,
                                                                 int64_t to_per_block) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    const int64_t b = blockIdx.z;
    if (c >= C) return;
    const int64_t to0 = (int64_t)blockIdx.x * to_per_block;
    const int64_t to1 = (to0 + to_per_block < To) ? to0 + to_per_block : To;
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0.f;
    float accb = 0.f;
    const float* zb = z + b * T * F * C + c;
    for (int64_t to = to0; to < to1; ++to) {
        const float* gz = du + ((b * To + to) * Fo) * C + c;
        for (int fo = 0; fo < Fo; ++fo) {
            const float g = gz[(int64_t)fo * C];
            accb += g;
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const int64_t t = 2 * to + dt - 1;
                if (t < 0 || t >= T) continue;
#pragma unroll
                for (int df = 0; df < 3; ++df) {
                    const int f = 2 * fo + df - 1;
                    if (f < 0 || f >= F) continue;
                    acc[dt * 3 + df] += g * silu_f(zb[(t * F + f) * C]);
                }
            }
        }
    }
    const int64_t tile = (int64_t)blockIdx.z * gridDim.x + blockIdx.x;
#pragma unroll
    for (int j = 0; j < 9; ++j) partial_w[(tile * C + c) * 9 + j] = acc[j];
    partial_b[tile * C + c] = accb;
}

template <bool FLIP>
int launch_dw1d(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t T, int64_t C, int64_t KW,
                float beta, hipStream_t st, int ngroups = 1, int64_t pstride = 0) {
    dim3 grid((unsigned)dyn::cdiv(T, TT), (unsigned)dyn::cdiv(C, 256), (unsigned)B), blk(256);
#define GO(K) hipLaunchKernelGGL((dwconv1d_kernel<K, FLIP>), grid, blk, 0, st, x, w, bias, y, T, (int)C, beta, ngroups, pstride)
    switch (KW) {
        case 3: GO(3); break; case 5: GO(5); break; case 7: GO(7); break; case 9: GO(9); break;
        case 15: GO(15); break; case 31: GO(31); break;
        default: dyn::set_error("dwconv1d: unsupported kernel width %lld", (long long)KW); return DYN_E_UNSUPPORTED;
    }
#undef GO
    return dyn::check_launch("dyn_dwconv1d");
}

// Rows (time steps) per wgrad workgroup: short serial chains and thousands of workgroups; the partial sums are
// combined afterwards by the fixed-order 2-D reducer (reduce.h).
inline int64_t wgrad_tiles(int64_t B, int64_t T, int64_t* per_block) {
    int64_t chunks = dyn::cdiv(T, 8);
    if (chunks > 2048) chunks = 2048;
    if (chunks < 1) chunks = 1;
    *per_block = dyn::cdiv(T > 0 ? T : 1, chunks);
    chunks = dyn::cdiv(T > 0 ? T : 1, *per_block);
    return chunks;
}

// 2-D subsampling weight gradients: one wave per tile, enough tiles (up to 4096 per sample) to fill the chip at B = 1.
inline int64_t wgrad_tiles2d(int64_t To, int64_t* per_block) {
    int64_t chunks = To < 4096 ? To : 4096;
    if (chunks < 1) chunks = 1;
    *per_block = dyn::cdiv(To > 0 ? To : 1, chunks);
    return dyn::cdiv(To > 0 ? To : 1, *per_block);
}

}  // namespace

extern "C" int dyn_dwconv1d_fwd(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t T, int64_t C,
                                int64_t KW, void* stream) {
    DYN_REQUIRE(x && w && y && B >= 0 && T >= 0 && C > 0, DYN_E_ARG, "dyn_dwconv1d_fwd: bad arguments");
    if (B == 0 || T == 0) return DYN_OK;
    return launch_dw1d<false>(x, w, bias, y, B, T, C, KW, 0.f, (hipStream_t)stream);
}

extern "C" int dyn_dwconv1d_dgrad(const float* dy, const float* w, float* dx, int64_t B, int64_t T, int64_t C, int64_t KW,
                                  float dx_beta, void* stream) {
    DYN_REQUIRE(dy && w && dx && B >= 0 && T >= 0 && C > 0, DYN_E_ARG, "dyn_dwconv1d_dgrad: bad arguments");
    if (B == 0 || T == 0) return DYN_OK;
    return launch_dw1d<true>(dy, w, nullptr, dx, B, T, C, KW, dx_beta, (hipStream_t)stream);
}

// lockstep-group variants (see dyn_layernorm_fwd_g, include/dyneval.h): sample b uses / accumulates into the filters of replica b % n_groups
extern "C" int dyn_dwconv1d_dgrad_g(const float* dy, const float* w, float* dx, int64_t B, int64_t T, int64_t C, int64_t KW, float dx_beta,
                                    int64_t n_groups, int64_t param_stride, void* stream) {
    DYN_REQUIRE(dy && w && dx && B >= 0 && T >= 0 && C > 0 && n_groups >= 1 && B % n_groups == 0, DYN_E_ARG, "dyn_dwconv1d_dgrad_g: bad arguments");
    if (B == 0 || T == 0) return DYN_OK;
    return launch_dw1d<true>(dy, w, nullptr, dx, B, T, C, KW, dx_beta, (hipStream_t)stream, (int)n_groups, param_stride);
}

extern "C" int64_t dyn_dwconv1d_wgrad_workspace_bytes(int64_t B, int64_t T, int64_t C, int64_t KW) {
    int64_t per;
    const int64_t tiles = wgrad_tiles(B, T, &per) * B;
    return tiles * C * (KW + 1) * (int64_t)sizeof(float);
}

static int dwconv1d_wgrad_impl(const float* x, const float* dy, float* dw, float* dbias, float beta, int64_t B, int64_t T, int64_t C, int64_t KW,
                              void* workspace, int64_t workspace_bytes, int64_t n_groups, int64_t pstride, void* stream);

extern "C" int dyn_dwconv1d_wgrad(const float* x, const float* dy, float* dw, float* dbias, float beta, int64_t B, int64_t T,
                                  int64_t C, int64_t KW, void* workspace, int64_t workspace_bytes, void* stream) {
    return dwconv1d_wgrad_impl(x, dy, dw, dbias, beta, B, T, C, KW, workspace, workspace_bytes, 1, 0, stream);
}

extern "C" int dyn_dwconv1d_wgrad_g(const float* x, const float* dy, float* dw, float* dbias, float beta, int64_t B, int64_t T, int64_t C,
                                    int64_t KW, int64_t n_groups, int64_t param_stride, void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(n_groups >= 1 && B % n_groups == 0, DYN_E_ARG, "dyn_dwconv1d_wgrad_g: the batch must hold whole chunks of n_groups samples");
    return dwconv1d_wgrad_impl(x, dy, dw, dbias, beta, B, T, C, KW, workspace, workspace_bytes, n_groups, param_stride, stream);
}

static int dwconv1d_wgrad_impl(const float* x, const float* dy, float* dw, float* dbias, float beta, int64_t B, int64_t T, int64_t C, int64_t KW,
                              void* workspace, int64_t workspace_bytes, int64_t n_groups, int64_t pstride, void* stream) {
    DYN_REQUIRE(x && dy && dw && B >= 0 && T >= 0 && C > 0, DYN_E_ARG, "dyn_dwconv1d_wgrad: bad arguments");
    if (B == 0 || T == 0) return DYN_OK;
    int64_t per;
    const int64_t chunks = wgrad_tiles(B, T, &per), tiles = chunks * B;
    DYN_REQUIRE(workspace && workspace_bytes >= tiles * C * (KW + 1) * (int64_t)sizeof(float), DYN_E_WORKSPACE,
                "dyn_dwconv1d_wgrad: workspace too small");
    float* pw = dyn::partials_alloc(workspace, tiles * C * (KW + 1) * (int64_t)sizeof(float));   // the workspace, or the open deferral context's arena
    float* pb = pw + tiles * C * KW;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)chunks, (unsigned)dyn::cdiv(C, 256), (unsigned)B), blk(256);
#define GO(K) hipLaunchKernelGGL((dwconv1d_wgrad_kernel<K>), grid, blk, 0, st, x, dy, pw, pb, T, (int)C, per)
    switch (KW) {
        case 3: GO(3); break; case 5: GO(5); break; case 7: GO(7); break; case 9: GO(9); break;
        case 15: GO(15); break; case 31: GO(31); break;
        default: dyn::set_error("dwconv1d_wgrad: unsupported kernel width %lld", (long long)KW); return DYN_E_UNSUPPORTED;
    }
#undef GO
    if (n_groups <= 1) {
        dyn::reduce_or_defer(pw, dw, tiles, C * KW, beta, st);
        if (dbias) dyn::reduce_or_defer(pb, dbias, tiles, C, beta, st);
    } else {        // a sample's tiles are contiguous (tile = b * chunks + chunk): one reduction per sample into its replica's gradient
        for (int64_t b = 0; b < B; ++b) {
            const int64_t po = (b % n_groups) * pstride;
            const float wb = b < n_groups ? beta : 1.f;
            dyn::reduce_or_defer(pw + b * chunks * C * KW, dw + po, chunks, C * KW, wb, st);
            if (dbias) dyn::reduce_or_defer(pb + b * chunks * C, dbias + po, chunks, C, wb, st);
        }
    }
    return dyn::check_launch("dyn_dwconv1d_wgrad");
}

extern "C" int dyn_conv2d_first_fwd(const float* x, const float* w, const float* bias, float* z, int64_t B, int64_t T,
                                    int64_t F, int64_t C, void* stream) {
    DYN_REQUIRE(x && w && bias && z && B >= 0 && T > 0 && F > 0 && C > 0, DYN_E_ARG, "dyn_conv2d_first_fwd: bad arguments");
    if (B == 0) return DYN_OK;
    const int64_t To = (T - 1) / 2 + 1, Fo = (F - 1) / 2 + 1;
    if (C % 4 == 0 && (((uintptr_t)z | (uintptr_t)bias) & 15) == 0) {
        const int64_t rows = B * To;
        hipLaunchKernelGGL(conv2d_first_fwd_v4_kernel, dim3((unsigned)dyn::cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, z,
                           T, (int)F, To, (int)Fo, (int)C, rows);
        return dyn::check_launch("dyn_conv2d_first_fwd");
    }
    dim3 grid((unsigned)To, (unsigned)dyn::cdiv(C, 256), (unsigned)B);
    hipLaunchKernelGGL(conv2d_first_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, z, T, (int)F, To, (int)Fo, (int)C);
    return dyn::check_launch("dyn_conv2d_first_fwd");
}

extern "C" int dyn_conv2d_first_dgrad(const float* dz, const float* w, float* dx, int64_t B, int64_t T, int64_t F, int64_t C,
                                      void* stream) {
    DYN_REQUIRE(dz && w && dx && B >= 0 && T > 0 && F > 0 && C > 0, DYN_E_ARG, "dyn_conv2d_first_dgrad: bad arguments");
    if (B == 0) return DYN_OK;
    const int64_t To = (T - 1) / 2 + 1, Fo = (F - 1) / 2 + 1;
    int64_t g = dyn::cdiv(B * T * F, 4);
    if (g > 65536) g = 65536;
    hipLaunchKernelGGL(conv2d_first_dgrad_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dz, w, dx, B, T, (int)F, To,
                       (int)Fo, (int)C);
    return dyn::check_launch("dyn_conv2d_first_dgrad");
}

extern "C" int64_t dyn_conv2d_wgrad_workspace_bytes(int64_t B, int64_t To, int64_t C) {
    int64_t per;
    const int64_t tiles = wgrad_tiles2d(To, &per) * B;
    return tiles * C * 10 * (int64_t)sizeof(float);
}

extern "C" int dyn_conv2d_first_wgrad(const float* x, const float* dz, float* dw, float* dbias, float beta, int64_t B,
                                      int64_t T, int64_t F, int64_t C, void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(x && dz && dw && dbias && B >= 0 && T > 0 && F > 0 && C > 0, DYN_E_ARG, "dyn_conv2d_first_wgrad: bad arguments");
    if (B == 0) return DYN_OK;
    const int64_t To = (T - 1) / 2 + 1, Fo = (F - 1) / 2 + 1;
    int64_t per;
    const int64_t chunks = wgrad_tiles2d(To, &per), tiles = chunks * B;
    DYN_REQUIRE(workspace && workspace_bytes >= tiles * C * 10 * (int64_t)sizeof(float), DYN_E_WORKSPACE,
                "dyn_conv2d_first_wgrad: workspace too small");
    float* pw = (float*)workspace;
    float* pb = pw + tiles * C * 9;
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0 && (((uintptr_t)dz) & 15) == 0) {
        hipLaunchKernelGGL((conv2d_s2_wgrad_v4_kernel<true>), dim3((unsigned)dyn::cdiv(tiles, 4)), dim3(256), 0, st, x, dz, pw, pb, T, (int)F,
                           To, (int)Fo, (int)C, per, chunks, tiles);
        dyn::ordered_before_launch(st);
        dyn::launch_reduce_partials_taps(pw, dw, dyn::cdiv(tiles, 4), C * 9, beta, (int)C, st);
        dyn::launch_reduce_partials(pb, dbias, dyn::cdiv(tiles, 4), C, beta, st);
        return dyn::check_launch("dyn_conv2d_first_wgrad");
    } else {
        dim3 grid((unsigned)chunks, (unsigned)dyn::cdiv(C, 256), (unsigned)B);
        hipLaunchKernelGGL(conv2d_first_wgrad_kernel, grid, dim3(256), 0, st, x, dz, pw, pb, T, (int)F, To, (int)Fo, (int)C, per);
    }
    dyn::ordered_before_launch(st);
    dyn::launch_reduce_partials(pw, dw, tiles, C * 9, beta, st);
    dyn::launch_reduce_partials(pb, dbias, tiles, C, beta, st);
    return dyn::check_launch("dyn_conv2d_first_wgrad");
}

// Fused first two subsampling stages (sub12_*_kernel above).  x [B, T, F] -> u2 [B, T2, F2, C], T1 = out_len(T), T2 = out_len(T1).
extern "C" int dyn_sub12_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* u2, int64_t B,
                             int64_t T, int64_t F, int64_t C, void* stream) {
    DYN_REQUIRE(x && w1 && b1 && w2 && b2 && u2 && B >= 0 && T > 0 && F > 0 && C > 0, DYN_E_ARG, "dyn_sub12_fwd: bad arguments");
    DYN_REQUIRE(C % 4 == 0 && ((((uintptr_t)u2) | ((uintptr_t)b1) | ((uintptr_t)b2)) & 15) == 0, DYN_E_UNSUPPORTED,
                "dyn_sub12_fwd: needs C %% 4 == 0 and 16-byte aligned u2 / biases (use the unfused kernels otherwise)");
    if (B == 0) return DYN_OK;
    const int64_t T1 = (T - 1) / 2 + 1, F1 = (F - 1) / 2 + 1, T2 = (T1 - 1) / 2 + 1, F2 = (F1 - 1) / 2 + 1;
    const int64_t rows = B * T2;
    hipLaunchKernelGGL(sub12_fwd_kernel, dim3((unsigned)dyn::cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, w1, b1, w2, b2, u2, T,
                       (int)F, T1, (int)F1, T2, (int)F2, (int)C, rows);
    return dyn::check_launch("dyn_sub12_fwd");
}

extern "C" int64_t dyn_sub12_bwd_workspace_bytes(int64_t B, int64_t T, int64_t C) {
    int64_t per;
    const int64_t T1 = (T - 1) / 2 + 1;
    const int64_t wgs = dyn::cdiv(wgrad_tiles2d(T1, &per) * B, 4);
    return wgs * C * 20 * (int64_t)sizeof(float);
}

// dw1 [C, 3, 3], db1 [C], dw2 [C, 3, 3], db2 [C] = beta * (.) + the gradients of both stages, given du2 [B, T2, F2, C].
extern "C" int dyn_sub12_bwd(const float* x, const float* du2, const float* w1, const float* b1, const float* w2, float* dw1, float* db1,
                             float* dw2, float* db2, float beta, int64_t B, int64_t T, int64_t F, int64_t C, void* workspace,
                             int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(x && du2 && w1 && b1 && w2 && dw1 && db1 && dw2 && db2 && B >= 0 && T > 0 && F > 0 && C > 0, DYN_E_ARG,
                "dyn_sub12_bwd: bad arguments");
    DYN_REQUIRE(C % 4 == 0 && ((((uintptr_t)du2) | ((uintptr_t)b1)) & 15) == 0, DYN_E_UNSUPPORTED,
                "dyn_sub12_bwd: needs C %% 4 == 0 and 16-byte aligned du2 / b1 (use the unfused kernels otherwise)");
    if (B == 0) return DYN_OK;
    const int64_t T1 = (T - 1) / 2 + 1, F1 = (F - 1) / 2 + 1, T2 = (T1 - 1) / 2 + 1, F2 = (F1 - 1) / 2 + 1;
    int64_t per;
    const int64_t chunks = wgrad_tiles2d(T1, &per), tiles = chunks * B, wgs = dyn::cdiv(tiles, 4);
    DYN_REQUIRE(workspace && workspace_bytes >= wgs * C * 20 * (int64_t)sizeof(float) && (((uintptr_t)workspace) & 15) == 0, DYN_E_WORKSPACE,
                "dyn_sub12_bwd: workspace too small");
    float* pw1 = dyn::partials_alloc(workspace, wgs * C * 20 * (int64_t)sizeof(float));   // the workspace, or the open deferral context's arena
    float* pb1 = pw1 + wgs * C * 9;
    float* pw2 = pb1 + wgs * C;
    float* pb2 = pw2 + wgs * C * 9;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sub12_bwd_kernel, dim3((unsigned)wgs), dim3(256), 0, st, x, du2, w1, b1, w2, pw1, pb1, pw2, pb2, T, (int)F, T1, (int)F1,
                       T2, (int)F2, (int)C, per, chunks, tiles);
    dyn::reduce_taps_or_defer(pw1, dw1, wgs, C * 9, beta, (int)C, st);
    dyn::reduce_or_defer(pb1, db1, wgs, C, beta, st);
    dyn::reduce_taps_or_defer(pw2, dw2, wgs, C * 9, beta, (int)C, st);
    dyn::reduce_or_defer(pb2, db2, wgs, C, beta, st);
    return dyn::check_launch("dyn_sub12_bwd");
}

extern "C" int dyn_dwconv2d_s2_fwd(const float* z, const float* w, const float* bias, float* u, int64_t B, int64_t T,
                                   int64_t F, int64_t C, void* stream) {
    DYN_REQUIRE(z && w && bias && u && B >= 0 && T > 0 && F > 0 && C > 0, DYN_E_ARG, "dyn_dwconv2d_s2_fwd: bad arguments");
    if (B == 0) return DYN_OK;
    const int64_t To = (T - 1) / 2 + 1, Fo = (F - 1) / 2 + 1;
    if (C % 4 == 0 && (((uintptr_t)z | (uintptr_t)u | (uintptr_t)bias) & 15) == 0) {
        const int64_t rows = B * To;
        hipLaunchKernelGGL(dwconv2d_s2_fwd_v4_kernel, dim3((unsigned)dyn::cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, z, w, bias, u,
                           T, (int)F, To, (int)Fo, (int)C, rows);
        return dyn::check_launch("dyn_dwconv2d_s2_fwd");
    }
    dim3 grid((unsigned)To, (unsigned)dyn::cdiv(C, 256), (unsigned)B);
    hipLaunchKernelGGL(dwconv2d_s2_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, z, w, bias, u, T, (int)F, To, (int)Fo, (int)C);
    return dyn::check_launch("dyn_dwconv2d_s2_fwd");
}

extern "C" int dyn_dwconv2d_s2_dgrad(const float* z, const float* w, const float* du, float* dz, int64_t B, int64_t T,
                                     int64_t F, int64_t C, void* stream) {
    DYN_REQUIRE(z && w && du && dz && B >= 0 && T > 0 && F > 0 && C > 0, DYN_E_ARG, "dyn_dwconv2d_s2_dgrad: bad arguments");
    if (B == 0) return DYN_OK;
    const int64_t To = (T - 1) / 2 + 1, Fo = (F - 1) / 2 + 1;
    if (C % 4 == 0 && ((((uintptr_t)z) | ((uintptr_t)du) | ((uintptr_t)dz)) & 15) == 0) {
        const int64_t rows = B * T;
        hipLaunchKernelGGL(dwconv2d_s2_dgrad_v4_kernel, dim3((unsigned)dyn::cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, z, w, du, dz,
                           T, (int)F, To, (int)Fo, (int)C, rows);
        return dyn::check_launch("dyn_dwconv2d_s2_dgrad");
    }
    dim3 grid((unsigned)T, (unsigned)dyn::cdiv(C, 256), (unsigned)B);
    hipLaunchKernelGGL(dwconv2d_s2_dgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, z, w, du, dz, T, (int)F, To, (int)Fo, (int)C);
    return dyn::check_launch("dyn_dwconv2d_s2_dgrad");
}

extern "C" int dyn_dwconv2d_s2_wgrad(const float* z, const float* du, float* dw, float* dbias, float beta, int64_t B,
                                     int64_t T, int64_t F, int64_t C, void* workspace, int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(z && du && dw && dbias && B >= 0 && T > 0 && F > 0 && C > 0, DYN_E_ARG, "dyn_dwconv2d_s2_wgrad: bad arguments");
    if (B == 0) return DYN_OK;
    const int64_t To = (T - 1) / 2 + 1, Fo = (F - 1) / 2 + 1;
    int64_t per;
    const int64_t chunks = wgrad_tiles2d(To, &per), tiles = chunks * B;
    DYN_REQUIRE(workspace && workspace_bytes >= tiles * C * 10 * (int64_t)sizeof(float), DYN_E_WORKSPACE,
                "dyn_dwconv2d_s2_wgrad: workspace too small");
    const bool v4 = C % 4 == 0 && ((((uintptr_t)z) | ((uintptr_t)du)) & 15) == 0;
    float* pw = v4 ? dyn::partials_alloc(workspace, tiles * C * 10 * (int64_t)sizeof(float)) : (float*)workspace;
    float* pb = pw + tiles * C * 9;
    hipStream_t st = (hipStream_t)stream;
    if (v4) {
        hipLaunchKernelGGL((conv2d_s2_wgrad_v4_kernel<false>), dim3((unsigned)dyn::cdiv(tiles, 4)), dim3(256), 0, st, z, du, pw, pb, T, (int)F,
                           To, (int)Fo, (int)C, per, chunks, tiles);
        dyn::reduce_taps_or_defer(pw, dw, dyn::cdiv(tiles, 4), C * 9, beta, (int)C, st);
        dyn::reduce_or_defer(pb, dbias, dyn::cdiv(tiles, 4), C, beta, st);
        return dyn::check_launch("dyn_dwconv2d_s2_wgrad");
    } else {
        dim3 grid((unsigned)chunks, (unsigned)dyn::cdiv(C, 256), (unsigned)B);
        hipLaunchKernelGGL(dwconv2d_s2_wgrad_kernel, grid, dim3(256), 0, st, z, du, pw, pb, T, (int)F, To, (int)Fo, (int)C, per);
    }
    dyn::ordered_before_launch(st);
    dyn::launch_reduce_partials(pw, dw, tiles, C * 9, beta, st);
    dyn::launch_reduce_partials(pb, dbias, tiles, C, beta, st);
    return dyn::check_launch("dyn_dwconv2d_s2_wgrad");
}
