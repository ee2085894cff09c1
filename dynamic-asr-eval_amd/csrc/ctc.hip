// CTC on the MI355X: greedy decode (argmax -> collapse repeats -> drop blank) and the log-sum-exp lattice
// (alpha/beta recursions, loss and gradient with respect to the log-probabilities).
//
// Replaces, in the reference's dynamic-eval loop:
//   * GreedyCTCDecoder(...)(out['final_posteriors'][-1].detach().cpu())   (reference lcasr/lib.py:498,559,565;
//     run_dynamic_eval_full.py:53,100) — here the [T, V+1] posteriors never leave HBM, only the ids do;
//   * torch.nn.CTCLoss(blank=V, reduction='sum') + its backward            (reference lcasr/lib.py:492,575,579;
//     reduction='mean' in wav2vec2/lib.py:351).  The lattice and the gradient follow torch's CPU kernel
//     (aten/src/ATen/native/LossCTC.cpp) operation by operation AND rounding by rounding: every exp / log goes through
//     libm_f32.h (glibc's expf / logf restated bit for bit), every add / subtract is a single fp32 operation in torch's
//     order, and the per-class log-sum of alpha + beta is accumulated PAIRWISE in descending lattice position exactly as
//     the CPU loop does (lcab = log(exp(lcab - max) + exp(ab - max)) + max).  On identical log-probs the outputs (nll, alpha,
//     beta, gradient) are bit-identical to torch._ctc_loss / its backward on the CPU
//     (tests/test_ops_gpu.py::test_ctc_lattice_is_bitwise_torch_cpu).
//
// MI355X mapping: the lattice is latency-bound (T serial steps), so everything that is NOT serial is pulled out
// of the scan and spread over the chip: the per-step gathers lp[t, label[s]] are pre-gathered by a full-grid kernel
// into a contiguous [T, L] slab (coalesced, prefetchable), the scan itself is one 1024-thread workgroup per sample
// with the previous lattice row in LDS and an 8-deep register prefetch of the slab, and the per-class reduction
// of alpha*beta is one workgroup per time step.  All reductions run in a fixed order (no float atomics).
#include "common.h"
#include "libm_f32.h"

namespace {

namespace glm = dyn::glm;

constexpr int SCAN_T = 1024;  // threads of the serial scan workgroup
template <int ITEMS> struct ScanCfg { static constexpr int PD = ITEMS >= 8 ? 2 : ITEMS >= 4 ? 4 : 8; };   // slab prefetch distance (time steps): 8 deep
// while it fits the 128 VGPRs of a 1024-thread workgroup; ITEMS x PD ring registers at ITEMS >= 4 would spill to scratch at 8

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

// ids[row] = argmax_c x[row, c]   (first maximum wins, as torch.argmax on CPU)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int32_t* __restrict__ ids,
                                                           int64_t rows, int C, int64_t ld, float* __restrict__ vals) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < rows; row += (int64_t)gridDim.x * 4) {
        const float* xr = x + row * ld;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float v = xr[c];
            if (better(v, c, bv, bi)) { bv = v; bi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) {
            ids[row] = bi == 0x7fffffff ? 0 : bi;
            if (vals) vals[row] = bv;
        }
    }
}

// Collapse repeats and drop blanks: one workgroup per sequence, ordered compaction by block prefix sums.
__global__ __launch_bounds__(1024) void ctc_collapse_kernel(const int32_t* __restrict__ ids, int32_t* __restrict__ out,
                                                             int32_t* __restrict__ out_len, int64_t T, int64_t stride,
                                                             int blank) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int64_t b = blockIdx.x;
    const int32_t* in = ids + b * stride;
    int32_t* o = out + b * stride;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < T; t0 += 1024) {
        const int64_t t = t0 + threadIdx.x;
        int id = -1, keep = 0;
        if (t < T) {
            id = in[t];
            const int prev = t > 0 ? in[t - 1] : -1;
            keep = (id != blank && id != prev) ? 1 : 0;
        }
        int incl = keep;  // inclusive wave scan
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
            const int n = __shfl_up(incl, o2, 64);
            if (lane >= o2) incl += n;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        const int b0 = base;
        if (keep) o[b0 + woff + incl - 1] = id;
        __syncthreads();
        if (threadIdx.x == 1023) base = b0 + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) out_len[b] = base;
}

// ---- lattice ------------------------------------------------------------------------------------------------

struct CtcDims {
    int64_t T_max, B, C;
    int64_t lp_st, lp_sb;  // element strides of log_probs for time and batch
    int64_t S_max;
    int64_t L_max;         // 2 * S_max + 1
    int blank;
};

// Per sample: prev_same[k] (previous target index with the same label, -1 if none) and is_last[k] (no later occurrence): the gradient
// kernel walks each label's occurrences from the last one down, the order of torch's CPU loop (s = 2S .. 0).
__global__ __launch_bounds__(1024) void ctc_prep_kernel(const int32_t* __restrict__ targets, const int32_t* __restrict__ tlen,
                                                        int32_t* __restrict__ prev_same, int32_t* __restrict__ is_last,
                                                        int64_t S_max) {
    // the target row is staged in LDS once: the two scans below are O(S^2) reads of it (S is a few hundred labels per window;
    // from global memory they cost 0.4 ms of pure L2 latency on the chain's critical path)
    extern __shared__ int32_t tg_s[];
    const int64_t b = blockIdx.x;
    const int S = tlen[b];
    const int32_t* tg = targets + b * S_max;
    for (int k = threadIdx.x; k < S; k += blockDim.x) tg_s[k] = tg[k];
    __syncthreads();
    // one wave per label position k, 64 candidate positions compared per step (ballot): with mostly distinct labels both searches run to
    // the end of the row, S / 64 steps instead of S dependent LDS reads per position
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int k = w; k < S; k += nw) {
        const int c = tg_s[k];
        int pv = -1;
        for (int j0 = k - 1; j0 >= 0; j0 -= 64) {   // 64 candidates below k per step, nearest first
            const int j = j0 - lane;
            const unsigned long long hit = __ballot(j >= 0 && tg_s[j] == c);
            if (hit) { pv = j0 - (__ffsll((long long)hit) - 1); break; }
        }
        int last = 1;
        for (int j0 = k + 1; j0 < S; j0 += 64) {
            const int j = j0 + lane;
            if (__ballot(j < S && tg_s[j] == c)) { last = 0; break; }
        }
        if (lane == 0) {
            prev_same[b * S_max + k] = pv;
            is_last[b * S_max + k] = last;
        }
    }
}

// slab[b][t][s] = lp[t, b, ext_label(s)]
__global__ __launch_bounds__(256) void ctc_gather_kernel(const float* __restrict__ lp, const int32_t* __restrict__ targets,
                                                          const int32_t* __restrict__ ilen, const int32_t* __restrict__ tlen,
                                                          float* __restrict__ slab, CtcDims d) {
    const int64_t t = blockIdx.x, b = blockIdx.y;
    if (t >= ilen[b]) return;
    const int L = 2 * tlen[b] + 1;
    const float* row = lp + t * d.lp_st + b * d.lp_sb;
    const int32_t* tg = targets + b * d.S_max;
    float* out = slab + (b * d.T_max + t) * d.L_max;
    for (int s = threadIdx.x; s < L; s += blockDim.x) out[s] = row[(s & 1) ? tg[s >> 1] : d.blank];
}

// log(exp(a - m) + exp(b - m) + exp(c - m)) + m with m = max(a, b, c) (0 when all three are -inf), LossCTC.cpp's expression
__device__ __forceinline__ float lse3(float a, float b, float c, const double* tab) {
    float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) m = 0.f;
    const float e = __fadd_rn(__fadd_rn(glm::exp_nonpos(__fsub_rn(a, m), tab), glm::exp_nonpos(__fsub_rn(b, m), tab)), glm::exp_nonpos(__fsub_rn(c, m), tab));
    return __fadd_rn(glm::logf_(e, tab), m);
}

// Forward scan: alpha[b][t][s]; nll[b].  One workgroup per sample, ITEMS lattice positions per thread.
template <int ITEMS>
__device__ __forceinline__ void ctc_alpha_body(float* rows, const double* tab, const float* __restrict__ slab, const int32_t* __restrict__ targets,
                                               const int32_t* __restrict__ ilen, const int32_t* __restrict__ tlen,
                                               float* __restrict__ alpha, float* __restrict__ nll, const CtcDims& d) {
    const int SCAN_T = blockDim.x;
    const int64_t b = blockIdx.x;
    const int T = ilen[b], S = tlen[b], L = 2 * S + 1;
    const int32_t* tg = targets + b * d.S_max;
    const float* sl = slab + b * d.T_max * d.L_max;
    float* al = alpha + b * d.T_max * d.L_max;
    float* prev = rows;
    float* cur = rows + d.L_max;
    constexpr int PD = ScanCfg<ITEMS>::PD;
    bool skip[ITEMS];
    float ring[ITEMS][PD];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int s = threadIdx.x + j * SCAN_T;
        skip[j] = (s < L) && (s & 1) && s >= 3 && tg[s >> 1] != tg[(s >> 1) - 1];
        float v = -INFINITY;
        if (s < L && T > 0) {
            if (s == 0) v = sl[0];
            else if (s == 1) v = sl[1];
            al[s] = v;
            prev[s] = v;
        }
#pragma unroll
        for (int u = 0; u < PD; ++u) ring[j][u] = (s < L && 1 + u < T) ? sl[(int64_t)(1 + u) * d.L_max + s] : 0.f;
    }
    __syncthreads();
    for (int t0 = 1; t0 < T; t0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int t = t0 + u;
            if (t < T) {  // uniform across the workgroup
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const int s = threadIdx.x + j * SCAN_T;
                    const float lpv = ring[j][u];
                    const int tn = t + PD;
                    ring[j][u] = (s < L && tn < T) ? sl[(int64_t)tn * d.L_max + s] : 0.f;
                    if (s < L) {
                        const float a = prev[s];
                        const float bb = s >= 1 ? prev[s - 1] : -INFINITY;
                        const float c = skip[j] ? prev[s - 2] : -INFINITY;
                        const float v = __fadd_rn(lse3(a, bb, c, tab), lpv);
                        cur[s] = v;
                        al[(int64_t)t * d.L_max + s] = v;
                    }
                }
                __syncthreads();
                float* tmp = prev; prev = cur; cur = tmp;
            }
        }
    }
    if (threadIdx.x == 0) {
        float r = INFINITY;
        if (T > 0) {   // LossCTC.cpp: -log(exp(l1 - m) + exp(l2 - m)) - m over the last two positions; an empty target has only position 0
            const float l1 = prev[L - 1];
            if (S == 0) r = -l1;
            else {
                const float l2 = prev[L - 2];
                float m = fmaxf(l1, l2);
                if (m == -INFINITY) m = 0.f;
                r = -__fadd_rn(glm::logf_(__fadd_rn(glm::exp_nonpos(__fsub_rn(l1, m), tab), glm::exp_nonpos(__fsub_rn(l2, m), tab)), tab), m);
            }
        }
        nll[b] = r;
    }
}

// Backward scan: beta recursion; overwrites alpha[b][t][s] with alpha + beta.
template <int ITEMS>
__device__ __forceinline__ void ctc_beta_body(float* rows, const double* tab, const float* __restrict__ slab, const int32_t* __restrict__ targets,
                                              const int32_t* __restrict__ ilen, const int32_t* __restrict__ tlen,
                                              float* __restrict__ alpha, const CtcDims& d) {
    const int SCAN_T = blockDim.x;
    const int64_t b = blockIdx.x;
    const int T = ilen[b], S = tlen[b], L = 2 * S + 1;
    if (T <= 0) return;
    const int32_t* tg = targets + b * d.S_max;
    const float* sl = slab + b * d.T_max * d.L_max;
    float* al = alpha + b * d.T_max * d.L_max;
    float* prev = rows;
    float* cur = rows + d.L_max;
    constexpr int PD = ScanCfg<ITEMS>::PD;
    bool skip[ITEMS];
    float ring[ITEMS][PD];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int s = threadIdx.x + j * SCAN_T;
        skip[j] = (s + 2 < L) && (s & 1) && tg[s >> 1] != tg[(s >> 1) + 1];
        if (s < L) {
            float v = -INFINITY;
            if (s == L - 1 || s == L - 2) v = sl[(int64_t)(T - 1) * d.L_max + s];
            prev[s] = v;
            al[(int64_t)(T - 1) * d.L_max + s] = v;
        }
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int t = T - 2 - u;
            ring[j][u] = (s < L && t >= 0) ? sl[(int64_t)t * d.L_max + s] : 0.f;
        }
    }
    __syncthreads();
    for (int t0 = T - 2; t0 >= 0; t0 -= PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int t = t0 - u;
            if (t >= 0) {
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const int s = threadIdx.x + j * SCAN_T;
                    const float lpv = ring[j][u];
                    const int tn = t - PD;
                    ring[j][u] = (s < L && tn >= 0) ? sl[(int64_t)tn * d.L_max + s] : 0.f;
                    if (s < L) {
                        const float a = prev[s];
                        const float bb = s + 1 < L ? prev[s + 1] : -INFINITY;
                        const float c = skip[j] ? prev[s + 2] : -INFINITY;
                        const float v = __fadd_rn(lse3(a, bb, c, tab), lpv);
                        cur[s] = v;
                        al[(int64_t)t * d.L_max + s] = v;
                    }
                }
                __syncthreads();
                float* tmp = prev; prev = cur; cur = tmp;
            }
        }
    }
}

// Both scans are independent until the gradient combines them, so they run as ONE launch: blockIdx.y = 0 walks alpha
// forward, blockIdx.y = 1 walks beta backward, each on its own CU (halves the serial latency of the CTC step).
template <int ITEMS>
__global__ __launch_bounds__(1024) void ctc_scan_kernel(const float* __restrict__ slab, const int32_t* __restrict__ targets,
                                                         const int32_t* __restrict__ ilen, const int32_t* __restrict__ tlen,
                                                         float* __restrict__ alpha, float* __restrict__ beta, float* __restrict__ nll,
                                                         CtcDims d) {
    extern __shared__ __attribute__((aligned(16))) double scan_lds[];  // the libm tables, then 2 * L_max floats
    glm::stage_table(scan_lds);   // the bodies barrier before their first lattice step
    float* rows = (float*)(scan_lds + glm::TABLE_DOUBLES);
    if (blockIdx.y == 0) ctc_alpha_body<ITEMS>(rows, scan_lds, slab, targets, ilen, tlen, alpha, nll, d);
    else ctc_beta_body<ITEMS>(rows, scan_lds, slab, targets, ilen, tlen, beta, d);
}

// grad[t, b, c] = (exp(lp) - exp(lcab[c] + nll - lp)) * g_b, lcab[c] = log sum_{s: label(s) = c} exp(alpha+beta)[t, s], with the sum taken
// the way LossCTC.cpp's CPU loop takes it: lattice positions in DESCENDING order, one pairwise log-add per position
// (lcab = ab if lcab == -inf else log(exp(lcab - max) + exp(ab - max)) + max).  One workgroup per (t, b): alpha + beta of the row goes to LDS,
// all classes get exp(lp) * g, then the classes that occur in the target are corrected: blank (positions 2S, 2S-2, .., 0) by ONE thread — the
// chain is serial by construction, S + 1 dependent log-adds, while the 2048 time steps of a window spread over the chip — and every other label by the
// thread that owns its LAST occurrence walking the prev_same chain.
__device__ __forceinline__ float log_add_pair(float lcab, float ab, const double* tab) {
    if (lcab == -INFINITY) return ab;
    const float mx = lcab < ab ? ab : lcab;   // std::max
    return __fadd_rn(glm::logf_(__fadd_rn(glm::exp_nonpos(__fsub_rn(lcab, mx), tab), glm::exp_nonpos(__fsub_rn(ab, mx), tab)), tab), mx);
}

__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ lp, const float* __restrict__ alpha_,
                                                        const float* __restrict__ beta_,
                                                        const int32_t* __restrict__ targets, const int32_t* __restrict__ prev_same,
                                                        const int32_t* __restrict__ is_last, const int32_t* __restrict__ ilen, const int32_t* __restrict__ tlen,
                                                        const float* __restrict__ nll, float* __restrict__ grad, int64_t g_st,
                                                        int64_t g_sb, float grad_scale, int mean_reduction, CtcDims d) {
    extern __shared__ __attribute__((aligned(16))) double grad_lds[];   // the libm tables, then alpha + beta of this row (L floats)
    const int64_t t = blockIdx.x, b = blockIdx.y;
    const int T = ilen[b], S = tlen[b], L = 2 * S + 1;
    float* gr = grad + t * g_st + b * g_sb;
    if (t >= T) {  // padded frames get zero gradient
        for (int c = threadIdx.x; c < d.C; c += blockDim.x) gr[c] = 0.f;
        return;
    }
    const double* tab = grad_lds;
    float* ab = (float*)(grad_lds + glm::TABLE_DOUBLES);
    glm::stage_table(grad_lds);
    float g = grad_scale;
    if (mean_reduction) g = (grad_scale / (float)d.B) / (float)(S > 0 ? S : 1);   // MeanBackward, then DivBackward by clamp_min(target_len, 1)
    const float* row = lp + t * d.lp_st + b * d.lp_sb;
    const float* ar = alpha_ + (b * d.T_max + t) * d.L_max;
    const float* br = beta_ + (b * d.T_max + t) * d.L_max;
    const int32_t* tg = targets + b * d.S_max;
    const int32_t* pv = prev_same + b * d.S_max;
    const float nl = nll[b];
    for (int s = threadIdx.x; s < L; s += blockDim.x) ab[s] = __fadd_rn(ar[s], br[s]);
    __syncthreads();
    // res = -inf for a class outside the target: (exp(lp) - exp(-inf)) * g
    for (int c = threadIdx.x; c < d.C; c += blockDim.x) gr[c] = __fmul_rn(glm::expf_(row[c], tab), g);
    __syncthreads();   // orders the stores above before the corrections below (same workgroup, same addresses)
    if (threadIdx.x == 0) {
        float lcab = -INFINITY;
        for (int k = S; k >= 0; --k) lcab = log_add_pair(lcab, ab[2 * k], tab);
        const float lpb = row[d.blank];
        gr[d.blank] = __fmul_rn(__fsub_rn(glm::expf_(lpb, tab), glm::expf_(__fsub_rn(__fadd_rn(lcab, nl), lpb), tab)), g);
    }
    // labels: a thread of waves 1 .. 3 handles target index k if it is the last occurrence of its label (wave 0 is busy with the blank chain)
    for (int k = (int)threadIdx.x - 64; k < S; k += (int)blockDim.x - 64) {
        if (k < 0 || !is_last[b * d.S_max + k]) continue;
        const int c = tg[k];
        float lcab = -INFINITY;
        for (int j = k; j >= 0; j = pv[j]) lcab = log_add_pair(lcab, ab[2 * j + 1], tab);
        const float lpc = row[c];
        gr[c] = __fmul_rn(__fsub_rn(glm::expf_(lpc, tab), glm::expf_(__fsub_rn(__fadd_rn(lcab, nl), lpc), tab)), g);
    }
}

__global__ void ctc_loss_reduce_kernel(const float* __restrict__ nll, const int32_t* __restrict__ tlen, float* loss, int64_t B,
                                       int mean_reduction) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = 0.f;
    for (int64_t b = 0; b < B; ++b) {
        const int S = tlen[b];
        s += mean_reduction ? nll[b] / (float)(S > 0 ? S : 1) : nll[b];
    }
    *loss = mean_reduction ? s / (float)B : s;
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct CtcWs {
    float* slab; float* alpha; float* beta; float* nll; int32_t* prev_same; int32_t* is_last;
    int64_t total;
};

CtcWs carve(void* ws, int64_t T, int64_t B, int64_t S_max) {
    const int64_t L = 2 * S_max + 1;
    CtcWs w;
    char* p = (char*)ws;
    int64_t off = 0;
    w.slab = (float*)(p + off); off += align_up(B * T * L * 4, 256);
    w.alpha = (float*)(p + off); off += align_up(B * T * L * 4, 256);
    w.beta = (float*)(p + off); off += align_up(B * T * L * 4, 256);
    w.nll = (float*)(p + off); off += align_up(B * 4, 256);
    w.prev_same = (int32_t*)(p + off); off += align_up((B * S_max + 1) * 4, 256);
    w.is_last = (int32_t*)(p + off); off += align_up((B * S_max + 1) * 4, 256);
    w.total = off;
    return w;
}

}  // namespace

extern "C" int dyn_ctc_greedy(const float* log_probs, int64_t B, int64_t T, int64_t C, int64_t ld, int32_t blank,
                              int32_t* argmax_ids, int32_t* out_ids, int32_t* out_len, void* stream) {
    DYN_REQUIRE(log_probs && argmax_ids && out_ids && out_len && B >= 0 && T >= 0 && C > 0 && ld >= C, DYN_E_ARG,
                "dyn_ctc_greedy: bad arguments");
    if (B == 0) return DYN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (T > 0) {
        int64_t g = dyn::cdiv(B * T, 4);
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)g), dim3(256), 0, st, log_probs, argmax_ids, B * T, (int)C, ld, (float*)nullptr);
    }
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3((unsigned)B), dim3(1024), 0, st, argmax_ids, out_ids, out_len, T, T, (int)blank);
    return dyn::check_launch("dyn_ctc_greedy");
}

extern "C" int dyn_argmax_rows(const float* x, int64_t rows, int64_t C, int64_t ld, int32_t* ids, float* vals, void* stream) {
    DYN_REQUIRE(x && ids && rows >= 0 && C > 0 && ld >= C, DYN_E_ARG, "dyn_argmax_rows: bad arguments");
    if (rows == 0) return DYN_OK;
    int64_t g = dyn::cdiv(rows, 4);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, ids, rows, (int)C, ld, vals);
    return dyn::check_launch("dyn_argmax_rows");
}

extern "C" int64_t dyn_ctc_loss_workspace_bytes(int64_t T, int64_t B, int64_t S_max) {
    return carve(nullptr, T, B, S_max > 0 ? S_max : 1).total;
}

extern "C" int dyn_ctc_loss_workspace_layout(int64_t T, int64_t B, int64_t S_max, int64_t* offsets4, int64_t* row_len) {
    DYN_REQUIRE(offsets4 && row_len && T > 0 && B > 0 && S_max >= 0, DYN_E_ARG, "dyn_ctc_loss_workspace_layout: bad arguments");
    const int64_t Sm = S_max > 0 ? S_max : 1;
    const CtcWs w = carve(nullptr, T, B, Sm);
    offsets4[0] = (char*)w.slab - (char*)nullptr;
    offsets4[1] = (char*)w.alpha - (char*)nullptr;
    offsets4[2] = (char*)w.beta - (char*)nullptr;
    offsets4[3] = (char*)w.nll - (char*)nullptr;
    *row_len = 2 * Sm + 1;
    return DYN_OK;
}

namespace {
__global__ __launch_bounds__(256) void libm_f32_kernel(const float* __restrict__ x, int64_t n, float* y_exp, float* y_log, float* y_np) {
    __shared__ double tab[glm::TABLE_DOUBLES];
    glm::stage_table(tab);
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (y_exp) y_exp[i] = glm::expf_(v, tab);
        if (y_log) y_log[i] = glm::logf_(v, tab);
        if (y_np) y_np[i] = glm::exp_nonpos(v, tab);
    }
}
}  // namespace

extern "C" int dyn_libm_f32(const float* x, int64_t n, float* y_exp, float* y_log, float* y_exp_nonpos, void* stream) {
    DYN_REQUIRE(x && n >= 0, DYN_E_ARG, "dyn_libm_f32: bad arguments");
    if (n == 0) return DYN_OK;
    int64_t g = dyn::cdiv(n, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(libm_f32_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, n, y_exp, y_log, y_exp_nonpos);
    return dyn::check_launch("dyn_libm_f32");
}

extern "C" int dyn_ctc_loss(const float* log_probs, int64_t T, int64_t B, int64_t C, int64_t lp_stride_t, int64_t lp_stride_b,
                            const int32_t* targets, int64_t S_max, const int32_t* input_lengths, const int32_t* target_lengths,
                            int32_t blank, int32_t reduction, float grad_scale, float* loss, float* nll_per_sample,
                            float* grad, int64_t g_stride_t, int64_t g_stride_b, void* workspace, int64_t workspace_bytes,
                            void* stream) {
    DYN_REQUIRE(log_probs && targets && input_lengths && target_lengths && loss && T > 0 && B > 0 && C > 0 && S_max >= 0 &&
                    blank >= 0 && blank < C && (reduction == 0 || reduction == 1),
                DYN_E_ARG, "dyn_ctc_loss: bad arguments");
    const int64_t Sm = S_max > 0 ? S_max : 1;
    const int64_t L = 2 * Sm + 1;
    // also bounds the dynamic LDS of ctc_prep_kernel (S_max * 4 bytes <= 16 KB) and of the scan (2 * L floats <= 64 KB)
    DYN_REQUIRE(L <= 8 * SCAN_T, DYN_E_UNSUPPORTED, "dyn_ctc_loss: lattice width %lld > %d unsupported", (long long)L, 8 * SCAN_T);
    CtcWs w = carve(workspace, T, B, Sm);
    DYN_REQUIRE(workspace && workspace_bytes >= w.total, DYN_E_WORKSPACE, "dyn_ctc_loss: workspace %lld < %lld bytes",
                (long long)workspace_bytes, (long long)w.total);
    CtcDims d;
    d.T_max = T; d.B = B; d.C = C; d.lp_st = lp_stride_t; d.lp_sb = lp_stride_b; d.S_max = Sm; d.L_max = L; d.blank = blank;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ctc_prep_kernel, dim3((unsigned)B), dim3(1024), (size_t)Sm * sizeof(int32_t), st, targets, target_lengths, w.prev_same, w.is_last, Sm);
    hipLaunchKernelGGL(ctc_gather_kernel, dim3((unsigned)T, (unsigned)B), dim3(256), 0, st, log_probs, targets, input_lengths,
                       target_lengths, w.slab, d);
    int threads = (int)((L + 63) / 64 * 64);
    if (threads > SCAN_T) threads = SCAN_T;
    const int items = (int)dyn::cdiv(L, threads);
    const size_t shm = (size_t)2 * L * sizeof(float) + glm::TABLE_DOUBLES * sizeof(double);
    const dim3 sgrid((unsigned)B, grad ? 2u : 1u);
#define GO_S(I) hipLaunchKernelGGL((ctc_scan_kernel<I>), sgrid, dim3(threads), shm, st, w.slab, targets, input_lengths, target_lengths, w.alpha, w.beta, w.nll, d)
    if (items <= 1) GO_S(1); else if (items <= 2) GO_S(2); else if (items <= 4) GO_S(4); else GO_S(8);
    hipLaunchKernelGGL(ctc_loss_reduce_kernel, dim3(1), dim3(64), 0, st, w.nll, target_lengths, loss, B, (int)reduction);
    if (nll_per_sample) {
        hipError_t e = hipMemcpyAsync(nll_per_sample, w.nll, B * sizeof(float), hipMemcpyDeviceToDevice, st);
        DYN_REQUIRE(e == hipSuccess, DYN_E_LAUNCH, "dyn_ctc_loss: copy of per-sample nll failed");
    }
    if (grad) {
        hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)T, (unsigned)B), dim3(256), (size_t)L * sizeof(float) + glm::TABLE_DOUBLES * sizeof(double), st, log_probs, w.alpha, w.beta, targets,
                           w.prev_same, w.is_last, input_lengths, target_lengths, w.nll, grad, g_stride_t, g_stride_b, grad_scale,
                           (int)reduction, d);
    }
#undef GO_S
    return dyn::check_launch("dyn_ctc_loss");
}
