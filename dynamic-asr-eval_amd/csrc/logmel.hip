// Log-mel front end on device: 16 kHz waveform -> [80, T] log-mel, per-bin normalised (100 frames / s).
// Replaces upstream `lcasr.utils.audio_tools.processing_chain / to_spectogram` as called by the dataset adapters
// (reference lcasr/earnings22/run.py:61, tedlium/run.py:94, chime6/run.py:61-68) — today a CPU step serial with the GPU.
//
// MI355X mapping: the STFT is ONE fp32-MFMA GEMM.  Frames are overlapping rows of the reflect-padded signal, so the
// frame matrix is never materialised: A[t, n] = xpad[t * hop + n] is addressed with lda = hop (160) < K (400) and the
// Hann window and the 56-sample centring offset of the 400-tap window inside the 512-point frame are folded into the
// DFT basis B[n, k] = w[n] * {cos, -sin}(2 pi k (n + 56) / 512)  (dyn_gemm_f32, NN).  The remaining kernels are
// HBM-bound single passes: power, mel projection (second GEMM), log + column moments, normalise + transpose.
#include "common.h"
#include "reduce.h"

namespace {
constexpr int TPB = 256;

// out[i] = x[reflect(i - pad)], i in [0, n + 2 pad): torch.stft(center=True, pad_mode='reflect')
__global__ __launch_bounds__(TPB) void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, int64_t pad) {
    const int64_t total = n + 2 * pad;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        int64_t j = i - pad;
        if (j < 0) j = -j;
        if (j >= n) j = 2 * (n - 1) - j;
        j = j < 0 ? 0 : (j >= n ? n - 1 : j);
        out[i] = x[j];
    }
}

// reim [T, 2*KP] (re in [0,KP), im in [KP, 2KP)) -> P [T, KP] = re^2 + im^2
__global__ __launch_bounds__(TPB) void power_kernel(const float* __restrict__ reim, float* __restrict__ P, int64_t T, int KP) {
    const int64_t total = T * KP;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int64_t t = i / KP;
        const int k = (int)(i % KP);
        const float re = reim[t * 2 * KP + k], im = reim[t * 2 * KP + KP + k];
        P[i] = re * re + im * im;
    }
}

// mel [T, F] -> log(mel + eps) in place; partial [block, 2F] = per-column (sum, sum of squares) of this block's rows
__global__ __launch_bounds__(TPB) void log_moments_kernel(float* __restrict__ mel, float* __restrict__ partial, int64_t T, int F,
                                                          float eps, int64_t rows_per_block) {
    __shared__ float red[2][TPB];
    const int f = threadIdx.x % F;
    const int lane_r = threadIdx.x / F;          // row lane inside the block
    const int rl = TPB / F;                      // row lanes (3 for F = 80)
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < T) ? r0 + rows_per_block : T;
    float s = 0.f, q = 0.f;
    if (lane_r < rl)
        for (int64_t t = r0 + lane_r; t < r1; t += rl) {
            const float v = logf(mel[t * F + f] + eps);
            mel[t * F + f] = v;
            s += v;
            q += v * v;
        }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < F) {
        float ts = 0.f, tq = 0.f;
        for (int k = 0; k < rl; ++k) { ts += red[0][k * F + threadIdx.x]; tq += red[1][k * F + threadIdx.x]; }
        partial[(int64_t)blockIdx.x * 2 * F + threadIdx.x] = ts;
        partial[(int64_t)blockIdx.x * 2 * F + F + threadIdx.x] = tq;
    }
}

// out[f, t] = (x[t, f] - mean_f) / std_f with the unbiased std over T (sums [2F] = column sum and sum of squares)
__global__ __launch_bounds__(TPB) void normalize_transpose_kernel(const float* __restrict__ x, const float* __restrict__ sums,
                                                                  float* __restrict__ out, int64_t T, int F, int normalize) {
    __shared__ float tile[32][33];
    const int64_t t0 = (int64_t)blockIdx.x * 32;
    const int f0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int64_t t = t0 + j;
        const int f = f0 + tx;
        tile[j][tx] = (t < T && f < F) ? x[t * F + f] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int f = f0 + j;
        const int64_t t = t0 + tx;
        if (f < F && t < T) {
            float v = tile[tx][j];
            if (normalize) {
                const double s = sums[f], q = sums[F + f];
                const double mean = s / (double)T;
                const double var = T > 1 ? (q - s * mean) / (double)(T - 1) : 1.0;
                v = (float)(((double)v - mean) / sqrt(var > 0.0 ? var : 1.0));
            }
            out[(int64_t)f * T + t] = v;
        }
    }
}
}  // namespace

// Per-row (mean, unbiased std) normalisation of a [R, T] spectrogram in place or out of place: the renormalisation after
// the CHiME-6 channel average, reference lcasr/chime6/run.py:66-68  (spec - spec.mean(-1)) / spec.std(-1).
// One 1024-thread workgroup per row; two passes (mean, then centred sum of squares) so long rows do not cancel.
namespace {
__global__ __launch_bounds__(1024) void rownorm_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t T) {
    __shared__ float red[16];
    const float* src = x + (int64_t)blockIdx.x * T;
    float* dst = out + (int64_t)blockIdx.x * T;
    float s = 0.f;
    for (int64_t t = threadIdx.x; t < T; t += 1024) s += src[t];
    const float mean = dyn::block_sum(s, red) / (float)T;
    float q = 0.f;
    for (int64_t t = threadIdx.x; t < T; t += 1024) { const float d = src[t] - mean; q += d * d; }
    const float var = dyn::block_sum(q, red) / (float)(T > 1 ? T - 1 : 1);
    const float inv = 1.f / sqrtf(var);
    for (int64_t t = threadIdx.x; t < T; t += 1024) dst[t] = (src[t] - mean) * inv;
}
}  // namespace

extern "C" int dyn_rownorm(const float* x, float* out, int64_t R, int64_t T, void* stream) {
    DYN_REQUIRE(x && out && R >= 0 && T >= 0 && R < (1ll << 31), DYN_E_ARG, "dyn_rownorm: bad arguments");
    if (R == 0 || T == 0) return DYN_OK;
    hipLaunchKernelGGL(rownorm_kernel, dim3((unsigned)R), dim3(1024), 0, (hipStream_t)stream, x, out, T);
    return dyn::check_launch("dyn_rownorm");
}

extern "C" int dyn_reflect_pad(const float* x, float* out, int64_t n, int64_t pad, void* stream) {
    DYN_REQUIRE(x && out && n > 1 && pad >= 0 && pad < n, DYN_E_ARG, "dyn_reflect_pad: bad arguments (need pad < n)");
    int64_t g = dyn::cdiv(n + 2 * pad, TPB);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)g), dim3(TPB), 0, (hipStream_t)stream, x, out, n, pad);
    return dyn::check_launch("dyn_reflect_pad");
}

extern "C" int dyn_stft_power(const float* reim, float* power, int64_t T, int64_t KP, void* stream) {
    DYN_REQUIRE(reim && power && T >= 0 && KP > 0, DYN_E_ARG, "dyn_stft_power: bad arguments");
    if (T == 0) return DYN_OK;
    int64_t g = dyn::cdiv(T * KP, TPB);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(power_kernel, dim3((unsigned)g), dim3(TPB), 0, (hipStream_t)stream, reim, power, T, (int)KP);
    return dyn::check_launch("dyn_stft_power");
}

extern "C" int64_t dyn_logmel_finish_workspace_bytes(int64_t T, int64_t F) {
    return (dyn::cdiv(T, 256) * 2 * F + 2 * F) * (int64_t)sizeof(float);
}

// mel [T, F] (overwritten with its log) -> out [F, T]; normalize != 0 applies the per-bin (mean, unbiased std) over T.
extern "C" int dyn_logmel_finish(float* mel, float* out, int64_t T, int64_t F, float eps, int32_t normalize, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(mel && out && T > 0 && F > 0 && F <= TPB, DYN_E_ARG, "dyn_logmel_finish: bad arguments");
    DYN_REQUIRE(workspace && workspace_bytes >= dyn_logmel_finish_workspace_bytes(T, F), DYN_E_WORKSPACE,
                "dyn_logmel_finish: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = dyn::cdiv(T, 256);
    float* partial = (float*)workspace;
    float* sums = partial + nb * 2 * F;
    hipLaunchKernelGGL(log_moments_kernel, dim3((unsigned)nb), dim3(TPB), 0, st, mel, partial, T, (int)F, eps, (int64_t)256);
    dyn::launch_reduce_partials(partial, sums, nb, 2 * F, 0.f, st);
    hipLaunchKernelGGL(normalize_transpose_kernel, dim3((unsigned)dyn::cdiv(T, 32), (unsigned)dyn::cdiv(F, 32)), dim3(TPB), 0, st, mel,
                       sums, out, T, (int)F, (int)normalize);
    return dyn::check_launch("dyn_logmel_finish");
}
