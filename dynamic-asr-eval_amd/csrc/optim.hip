// Fused adaptation step over a FLAT parameter buffer: one launch updates every parameter of the model.
//   MADGRAD  — the reference's default optimiser (reference lcasr/lib.py:14,458,494,581; run_half_concat_eval.py:98)
//   Adam     — reference nvidia_ctc/lib.py:43,155-160 (torch.optim.Adam semantics)
//   grad-norm clip — torch.nn.utils.clip_grad_norm_(…, 10.0) in reference wav2vec2/lib.py:442
// HBM-bound: MADGRAD reads p, g, s, nu, x0 and writes p, s, nu = 32 B / parameter; Adam 28 B / parameter.
// 16-B vector accesses, grid-stride over 256 workgroups (see grid_for).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TPB = 256;

// 256 workgroups (one per CU, 4 waves) instead of a chip-filling 2048: alone the step streams 13 % slower (0.50 -> 0.57 ms for 86 M
// parameters), but a launch that holds every wave slot keeps the other chains' matrix kernels from starting at all, while one workgroup
// per CU leaves them room (scripts/probe_overlap_mixed.py: a GEMM stream beside this stream, overlap gain 1.15 -> 1.33;
// job 799.2 -> 802.0 audio-s/s, alternating runs).  DYN_OPTIM_GRID overrides (A/B).
inline unsigned grid_for(int64_t n4) {
    int64_t g = dyn::cdiv(n4, TPB);
    static const int cap = [] { const char* e = getenv("DYN_OPTIM_GRID"); return e && atoi(e) > 0 ? atoi(e) : 256; }();
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

struct MadgradArgs {
    float lamb;      // (lr + eps) * sqrt(k + 1)
    float ck;        // 1 - momentum
    float eps;
    float weight_decay;
    int first_step;  // k == 0: s = nu = 0 and x0 = p are initialised in-kernel
};

__device__ __forceinline__ void madgrad_one(float& p, float g, float& s, float& nu, float& x0, const MadgradArgs a) {
    if (a.first_step) { s = 0.f; nu = 0.f; x0 = p; }
    if (a.weight_decay != 0.f) g += a.weight_decay * p;
    if (a.ck == 1.f) {  // momentum == 0: x0 is re-derived from the current iterate
        const float rms0 = cbrtf(nu) + a.eps;
        x0 = p + s / rms0;
    }
    nu += a.lamb * g * g;
    const float rms = cbrtf(nu) + a.eps;
    s += a.lamb * g;
    const float z = x0 - s / rms;
    p = (a.ck == 1.f) ? z : (1.f - a.ck) * p + a.ck * z;
}

__global__ __launch_bounds__(TPB) void madgrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ s,
                                                      float* __restrict__ nu, float* __restrict__ x0, int64_t n,
                                                      const MadgradArgs a) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 sv, nv, xv;
        if (a.first_step) {
            sv = nv = xv = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            sv = reinterpret_cast<float4*>(s)[i];
            nv = reinterpret_cast<float4*>(nu)[i];
            xv = reinterpret_cast<float4*>(x0)[i];
        }
        madgrad_one(pv.x, gv.x, sv.x, nv.x, xv.x, a);
        madgrad_one(pv.y, gv.y, sv.y, nv.y, xv.y, a);
        madgrad_one(pv.z, gv.z, sv.z, nv.z, xv.z, a);
        madgrad_one(pv.w, gv.w, sv.w, nv.w, xv.w, a);
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(s)[i] = sv;
        reinterpret_cast<float4*>(nu)[i] = nv;
        if (a.first_step || a.ck == 1.f) reinterpret_cast<float4*>(x0)[i] = xv;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        float pv = p[i], sv = s[i], nv = nu[i], xv = x0[i];
        madgrad_one(pv, g[i], sv, nv, xv, a);
        p[i] = pv; s[i] = sv; nu[i] = nv; x0[i] = xv;
    }
}

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay;
    float bc1, bc2_sqrt;  // 1 - beta1^t, sqrt(1 - beta2^t)
    int first_step;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs a) {
    if (a.first_step) { m = 0.f; v = 0.f; }
    if (a.weight_decay != 0.f) g += a.weight_decay * p;
    m = a.beta1 * m + (1.f - a.beta1) * g;     // exp_avg.lerp_(grad, 1 - beta1)
    v = a.beta2 * v + (1.f - a.beta2) * g * g;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= (a.lr / a.bc1) * (m / denom);
}

__global__ __launch_bounds__(TPB) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, const AdamArgs a) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = a.first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(m)[i];
        float4 vv = a.first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(v)[i];
        adam_one(pv.x, gv.x, mv.x, vv.x, a);
        adam_one(pv.y, gv.y, mv.y, vv.y, a);
        adam_one(pv.z, gv.z, mv.z, vv.z, a);
        adam_one(pv.w, gv.w, mv.w, vv.w, a);
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        float pv = p[i], mv = m[i], vv = v[i];
        adam_one(pv, g[i], mv, vv, a);
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}

// Sum of squares, stage 1: partial[block] (fixed-order block tree).  Stage 2 combines the partials in order and
// writes the clip coefficient min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_ semantics).
__global__ __launch_bounds__(TPB) void sumsq_partial_kernel(const float* __restrict__ g, float* __restrict__ partial, int64_t n) {
    __shared__ float red[8];
    float s = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) s += g[i] * g[i];
    s = dyn::block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void clip_coef_kernel(const float* __restrict__ partial, int nblocks, float max_norm, float* __restrict__ out2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < nblocks; ++i) s += (double)partial[i];
    const float norm = (float)sqrt(s);
    float coef = max_norm / (norm + 1e-6f);
    if (coef > 1.f) coef = 1.f;
    out2[0] = norm;
    out2[1] = coef;
}

__global__ __launch_bounds__(TPB) void scale_by_device_kernel(float* __restrict__ g, const float* __restrict__ coef, int64_t n) {
    const float c = coef[1];
    if (c == 1.f) return;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
        float4 v = reinterpret_cast<float4*>(g)[i];
        v.x *= c; v.y *= c; v.z *= c; v.w *= c;
        reinterpret_cast<float4*>(g)[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) g[i] *= c;
}

}  // namespace

extern "C" int dyn_madgrad_step(float* params, const float* grads, float* grad_sum, float* grad_sum_sq, float* x0, int64_t n,
                                float lr, float momentum, float weight_decay, float eps, int64_t step, void* stream) {
    DYN_REQUIRE(params && grads && grad_sum && grad_sum_sq && x0 && n >= 0 && step >= 0 && momentum >= 0.f && momentum < 1.f,
                DYN_E_ARG, "dyn_madgrad_step: bad arguments");
    if (n == 0) return DYN_OK;
    MadgradArgs a;
    // facebookresearch/madgrad: `lr = group["lr"] + eps; lamb = lr * (k + 1) ** 0.5`
    a.lamb = (float)(((double)lr + (double)eps) * sqrt((double)(step + 1)));
    a.ck = 1.f - momentum;
    a.eps = eps;
    a.weight_decay = weight_decay;
    a.first_step = step == 0;
    hipLaunchKernelGGL(madgrad_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, (hipStream_t)stream, params, grads, grad_sum,
                       grad_sum_sq, x0, n, a);
    return dyn::check_launch("dyn_madgrad_step");
}

extern "C" int dyn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream) {
    DYN_REQUIRE(params && grads && exp_avg && exp_avg_sq && n >= 0 && step >= 0, DYN_E_ARG, "dyn_adam_step: bad arguments");
    if (n == 0) return DYN_OK;
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    const double t = (double)(step + 1);
    a.bc1 = (float)(1.0 - pow((double)beta1, t));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, t));
    a.first_step = step == 0;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, n, a);
    return dyn::check_launch("dyn_adam_step");
}

extern "C" int64_t dyn_clip_grad_norm_workspace_bytes(int64_t n) { return (1024 + 4) * (int64_t)sizeof(float); }

// norm_and_coef[0] = total L2 norm, [1] = applied coefficient (device memory, 2 floats)
extern "C" int dyn_clip_grad_norm(float* grads, int64_t n, float max_norm, float* norm_and_coef, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    DYN_REQUIRE(grads && norm_and_coef && n >= 0, DYN_E_ARG, "dyn_clip_grad_norm: bad arguments");
    DYN_REQUIRE(workspace && workspace_bytes >= 1024 * (int64_t)sizeof(float), DYN_E_WORKSPACE, "dyn_clip_grad_norm: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)dyn::cdiv(n / 4 + 1, TPB);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(TPB), 0, st, grads, (float*)workspace, n);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, nb, max_norm, norm_and_coef);
    hipLaunchKernelGGL(scale_by_device_kernel, dim3(grid_for(n / 4 + 1)), dim3(TPB), 0, st, grads, norm_and_coef, n);
    return dyn::check_launch("dyn_clip_grad_norm");
}
