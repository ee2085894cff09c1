// Fused conformer conv-module core: GLU -> depthwise Conv1d (k = 9, 'same') -> RMSNorm / LayerNorm over channels -> SiLU
// in ONE pass over the pointwise-conv output (reference: upstream SCConformerXL conv module reached through
// model(audio_signal=...), reference lcasr/lib.py:550; `conv_kernel_size: 9`, earnings_finetune/lcasr160rb1.yaml:15).
//   u [B, T, 2C]  ->  s [B, T, C] = SiLU(norm_C(bias + sum_j w[c, j] * GLU(u)[t + j - 4, c]) * gamma (+ beta))
// Unfused this is four HBM-bound launches (GLU, dwconv, norm, SiLU) and three round trips of a [B, T, C] tensor.
// One 256-thread workgroup owns TT = 4 consecutive frames of one sample and ALL channels (C = 256 * NV, a thread owns
// channels tid, tid + 256, ...: coalesced), slides the GLU window through registers (halo frames are recomputed, never
// re-read from a neighbour), and finishes the per-frame channel statistics with one wavefront + LDS reduction for the
// whole tile.  With `save` the GLU output, the conv output and rstd/mean are also written for the (unfused) backward.
#include "common.h"

namespace {
constexpr int TT = 4, KW = 9, P = 4, TPB = 256;   // TT = 4: 1024 workgroups at B = 2, T = 2048 (16 gave 256 = one per CU: latency-bound, 1.4 TB/s)

template <int NV, bool LAYERNORM>
__global__ __launch_bounds__(TPB) void convmod_fwd_kernel(const float* __restrict__ u, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ s_out,
                                                          float* __restrict__ g_out, float* __restrict__ c_out, float* __restrict__ nn_out,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t T,
                                                          float eps, int ngroups, int64_t pstride) {
    constexpr int C = NV * 256;
    __shared__ float red[2][4][TT];
    __shared__ float stat[2][TT];
    const int64_t b = blockIdx.y;
    {   // lockstep group: sample b takes the parameters of replica b % ngroups, pstride elements apart (ngroups = 1: plain launch)
        const int64_t poff = (int64_t)(blockIdx.y % ngroups) * pstride;
        w += poff; gamma += poff;
        if (bias) bias += poff;
        if (beta) beta += poff;
    }
    const int64_t t0 = (int64_t)blockIdx.x * TT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float cv[NV][TT];
    float s1[TT], s2[TT];
#pragma unroll
    for (int k = 0; k < TT; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x + 256 * j;
        float wk[KW];
#pragma unroll
        for (int q = 0; q < KW; ++q) wk[q] = w[c * KW + q];
        const float bv = bias ? bias[c] : 0.f;
        float win[TT + 2 * P];
#pragma unroll
        for (int q = 0; q < TT + 2 * P; ++q) {
            const int64_t t = t0 + q - P;
            float g = 0.f;
            if (t >= 0 && t < T) {
                const float* ur = u + (b * T + t) * (2 * C);
                g = ur[c] * dyn::sigmoidf_(ur[C + c]);
                if (g_out && q >= P && q < TT + P) g_out[(b * T + t) * C + c] = g;
            }
            win[q] = g;
        }
#pragma unroll
        for (int k = 0; k < TT; ++k) {
            float acc = bv;
#pragma unroll
            for (int q = 0; q < KW; ++q) acc += wk[q] * win[k + q];
            cv[j][k] = acc;
            s1[k] += acc;
            s2[k] += acc * acc;
        }
    }
    // channel statistics of the TT frames: wavefront sums, then the 4 waves through LDS
#pragma unroll
    for (int k = 0; k < TT; ++k) {
        const float a = dyn::wave_sum(s1[k]), q = dyn::wave_sum(s2[k]);
        if (lane == 0) { red[0][wv][k] = a; red[1][wv][k] = q; }
    }
    __syncthreads();
    if (threadIdx.x < TT) {
        const int k = threadIdx.x;
        const float a = (red[0][0][k] + red[0][1][k]) + (red[0][2][k] + red[0][3][k]);
        const float q = (red[1][0][k] + red[1][1][k]) + (red[1][2][k] + red[1][3][k]);
        const float mean = LAYERNORM ? a / C : 0.f;
        const float var = LAYERNORM ? fmaxf(q / C - mean * mean, 0.f) : q / C;
        const float rstd = rsqrtf(var + eps);
        stat[0][k] = mean;
        stat[1][k] = rstd;
        const int64_t t = t0 + k;
        if (t < T) {
            if (rstd_out) rstd_out[b * T + t] = rstd;
            if (LAYERNORM && mean_out) mean_out[b * T + t] = mean;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x + 256 * j;
        const float gm = gamma[c], bt = (LAYERNORM && beta) ? beta[c] : 0.f;
#pragma unroll
        for (int k = 0; k < TT; ++k) {
            const int64_t t = t0 + k;
            if (t < T) {
                const float nn = (cv[j][k] - stat[0][k]) * stat[1][k] * gm + bt;
                s_out[(b * T + t) * C + c] = nn * dyn::sigmoidf_(nn);
                if (c_out) c_out[(b * T + t) * C + c] = cv[j][k];
                if (nn_out) nn_out[(b * T + t) * C + c] = nn;
            }
        }
    }
}
}  // namespace

// layernorm == 0: RMSNorm (beta ignored, mean_out unused); g_out / c_out / nn_out (= norm output) / mean_out / rstd_out may
// be NULL (no-grad pass).
static int convmod_launch(const float* u, const float* w, const float* bias, const float* gamma, const float* beta, float* s,
                          float* g_out, float* c_out, float* nn_out, float* mean_out, float* rstd_out, int64_t B, int64_t T, int64_t C,
                          int64_t KWIDTH, int32_t layernorm, float eps, int ngroups, int64_t pstride, void* stream);

extern "C" int dyn_convmod_fwd(const float* u, const float* w, const float* bias, const float* gamma, const float* beta, float* s,
                               float* g_out, float* c_out, float* nn_out, float* mean_out, float* rstd_out, int64_t B, int64_t T, int64_t C,
                               int64_t KWIDTH, int32_t layernorm, float eps, void* stream) {
    return convmod_launch(u, w, bias, gamma, beta, s, g_out, c_out, nn_out, mean_out, rstd_out, B, T, C, KWIDTH, layernorm, eps, 1, 0, stream);
}

// lockstep-group variant: sample b of the batch uses the parameters of replica b % n_groups, param_stride elements apart
extern "C" int dyn_convmod_fwd_g(const float* u, const float* w, const float* bias, const float* gamma, const float* beta, float* s,
                                 float* g_out, float* c_out, float* nn_out, float* mean_out, float* rstd_out, int64_t B, int64_t T, int64_t C,
                                 int64_t KWIDTH, int32_t layernorm, float eps, int64_t n_groups, int64_t param_stride, void* stream) {
    DYN_REQUIRE(n_groups >= 1 && B % n_groups == 0, DYN_E_ARG, "dyn_convmod_fwd_g: the batch must hold whole chunks of n_groups samples");
    return convmod_launch(u, w, bias, gamma, beta, s, g_out, c_out, nn_out, mean_out, rstd_out, B, T, C, KWIDTH, layernorm, eps, (int)n_groups,
                          param_stride, stream);
}

static int convmod_launch(const float* u, const float* w, const float* bias, const float* gamma, const float* beta, float* s,
                          float* g_out, float* c_out, float* nn_out, float* mean_out, float* rstd_out, int64_t B, int64_t T, int64_t C,
                          int64_t KWIDTH, int32_t layernorm, float eps, int ngroups, int64_t pstride, void* stream) {
    DYN_REQUIRE(u && w && gamma && s && B >= 0 && T >= 0 && C > 0, DYN_E_ARG, "dyn_convmod_fwd: bad arguments");
    DYN_REQUIRE(KWIDTH == KW && C % 256 == 0 && C <= 1024, DYN_E_UNSUPPORTED, "dyn_convmod_fwd: needs kernel width 9 and C in {256,512,768,1024}");
    if (B == 0 || T == 0) return DYN_OK;
    dim3 grid((unsigned)dyn::cdiv(T, TT), (unsigned)B), blk(TPB);
    hipStream_t st = (hipStream_t)stream;
#define GO(NV, LN) hipLaunchKernelGGL((convmod_fwd_kernel<NV, LN>), grid, blk, 0, st, u, w, bias, gamma, beta, s, g_out, c_out, nn_out, mean_out, rstd_out, T, eps, ngroups, pstride)
    const int nv = (int)(C / 256);
    if (layernorm) { if (nv == 1) GO(1, true); else if (nv == 2) GO(2, true); else if (nv == 3) GO(3, true); else GO(4, true); }
    else { if (nv == 1) GO(1, false); else if (nv == 2) GO(2, false); else if (nv == 3) GO(3, false); else GO(4, false); }
#undef GO
    return dyn::check_launch("dyn_convmod_fwd");
}
