// SpecAugment masks with the (start, width) pairs passed BY VALUE as kernel arguments (<= 32 masks): no device buffer, no
// host-to-device copy, hence no stream-ordered blocking memcpy in the per-window loop (a pageable H2D copy makes the host
// wait for everything queued before it — it would serialise the host with the previous window's backward).
// Same semantics as dyn_specaug_freqmask / dyn_specaug_timemask (reference lcasr/lib.py:499,541).
#include "common.h"

namespace {
constexpr int TPB = 256, MAXM = 32;
struct Masks { int32_t start[MAXM]; int32_t width[MAXM]; int n; };

__global__ __launch_bounds__(TPB) void freq_mask_args_kernel(float* x, int F, int64_t T, const Masks m, float value,
                                                             const float* __restrict__ value_dev) {
    if (value_dev) value = *value_dev;
    const int f = blockIdx.y;
    bool hit = false;
    for (int k = 0; k < m.n; ++k) hit |= (f >= m.start[k] && f < m.start[k] + m.width[k]);
    if (!hit) return;
    for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < T; t += (int64_t)gridDim.x * TPB) x[(int64_t)f * T + t] = value;
}

__global__ __launch_bounds__(TPB) void time_mask_args_kernel(float* x, int F, int64_t T, const Masks m, float value,
                                                             const float* __restrict__ value_dev) {
    if (value_dev) value = *value_dev;
    const int k = blockIdx.y;
    const int64_t a = m.start[k], wd = m.width[k];
    const int64_t total = (int64_t)F * wd;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
        const int64_t f = i / wd, t = a + i % wd;
        if (t >= 0 && t < T) x[f * T + t] = value;
    }
}
}  // namespace

// start_host / width_host are HOST arrays of n_masks <= 32 entries, read at launch time.
extern "C" int dyn_specaug_mask_args(float* x, int64_t F, int64_t T, const int32_t* start_host, const int32_t* width_host,
                                     int64_t n_masks, int32_t along_time, float value, const float* value_dev, void* stream) {
    DYN_REQUIRE(x && F > 0 && T >= 0 && n_masks >= 0 && n_masks <= MAXM && (n_masks == 0 || (start_host && width_host)), DYN_E_ARG,
                "dyn_specaug_mask_args: bad arguments (at most %d masks)", MAXM);
    if (T == 0 || n_masks == 0) return DYN_OK;
    Masks m;
    m.n = (int)n_masks;
    for (int k = 0; k < MAXM; ++k) { m.start[k] = k < n_masks ? start_host[k] : 0; m.width[k] = k < n_masks ? width_host[k] : 0; }
    hipStream_t st = (hipStream_t)stream;
    if (along_time) {
        hipLaunchKernelGGL(time_mask_args_kernel, dim3(64, (unsigned)n_masks), dim3(TPB), 0, st, x, (int)F, T, m, value, value_dev);
    } else {
        int64_t gx = dyn::cdiv(T, TPB * 4);
        if (gx > 256) gx = 256;
        hipLaunchKernelGGL(freq_mask_args_kernel, dim3((unsigned)gx, (unsigned)F), dim3(TPB), 0, st, x, (int)F, T, m, value, value_dev);
    }
    return dyn::check_launch("dyn_specaug_mask_args");
}
