// Error plumbing + identity queries of the C-ABI (see include/dyneval.h).
#include "common.h"
#include <string.h>

namespace dyn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace dyn

extern "C" const char* dyn_last_error(void) { return dyn::g_err; }
extern "C" const char* dyn_version(void) { return "dyneval-hip 0.1"; }
extern "C" const char* dyn_arch(void) { return "gfx950"; }

// A HIP stream whose kernels may only run on the compute units set in `mask` (bit i of word i / 32 = CU i; hipExtStreamCreateWithCUMask).
// The recording chains of lib.dynamic_eval_many can be given such streams (DYN_CHAIN_CU_MASK), so that a chain's matrix kernels leave a
// few CUs to the short HBM- / latency-bound kernels of the other chains.  Scheduling aid only; the caller owns the stream and destroys it.
extern "C" int dyn_stream_create_cu_mask(const uint32_t* mask, int32_t n_words, void** stream_out) {
    DYN_REQUIRE(mask && n_words > 0 && stream_out, DYN_E_ARG, "dyn_stream_create_cu_mask: bad arguments");
    hipStream_t st = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask);
    DYN_REQUIRE(e == hipSuccess, DYN_E_LAUNCH, "dyn_stream_create_cu_mask: %s", hipGetErrorString(e));
    *stream_out = (void*)st;
    return DYN_OK;
}

extern "C" int dyn_stream_destroy(void* stream) {
    DYN_REQUIRE(stream, DYN_E_ARG, "dyn_stream_destroy: null stream");
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    DYN_REQUIRE(e == hipSuccess, DYN_E_LAUNCH, "dyn_stream_destroy: %s", hipGetErrorString(e));
    return DYN_OK;
}
