// Error plumbing + identity queries of the C-ABI (see include/dyneval.h).
#include "common.h"
#include "reduce.h"
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

// ---- deferred column reductions -------------------------------------------------------------------------------------------------
// The reductions that end a backward kernel (LayerNorm / RMSNorm weight gradients, bias column sums) are launch-bound: ~7 us each for
// a few hundred KB, ~55 per window of the adapt step, and (profiles/r03_trace_overlap_chains3.json) alone on the chip while they run.
// Nothing reads their outputs before the optimiser, so between begin and flush they are recorded and then run as ONE launch per 96:
// item e of the table is reduced by the workgroups (., e) with the arithmetic of reduce_partials_2d_kernel (16 row-lanes, lane order),
// and items that accumulate into the SAME output (a shared parameter: the self-conditioning head's norm, six times per backward) form a
// chain walked by the workgroups of its first item, value carried in a register — the same fp32 operations in the same order as the
// separate launches: bit-identical.
namespace dyn {
namespace {
struct ReduceItem {
    const float* partial;
    float* out;
    int32_t P, n;
    float beta;
    int32_t next;    // next item of the same output (-1: none); >= 0 only on chain members
    int32_t head;    // 1: first item of its chain (its workgroups walk the chain), 0: handled by the head's workgroups
    int32_t taps_c;  // > 0: partial columns are laid out [tap][C], the output [C][taps] (reduce_partials_2d_taps_kernel); 0: same order
};
constexpr int kTable = 96;        // 96 * 40 B = 3.8 KB of kernel arguments
struct ReduceTable { ReduceItem it[kTable]; };
constexpr int kMaxItems = 1024;

struct DeferState {
    bool active = false;
    char* arena = nullptr;
    int64_t cap = 0, off = 0;
    int n = 0;
    hipStream_t stream = nullptr;   // the stream of the first recorded item: producers, later items and the flush must all use it
    bool have_stream = false;
    ReduceItem items[kMaxItems];
};
thread_local DeferState g_defer;

__global__ __launch_bounds__(1024) void reduce_partials_batched_kernel(const ReduceTable t) {
    __shared__ float red[16][64];
    int e = blockIdx.y;
    if (!t.it[e].head) return;
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t n = t.it[e].n;
    const int64_t col = (int64_t)blockIdx.x * 64 + cl;
    if ((int64_t)blockIdx.x * 64 >= n) return;
    float* out = t.it[e].out;
    const int64_t tc = t.it[e].taps_c;
    const int64_t ocol = tc > 0 ? (col % tc) * (n / tc) + col / tc : col;      // the same for every member of a chain
    float carried = 0.f;
    bool first = true;
    while (e >= 0) {
        const float* __restrict__ partial = t.it[e].partial;
        const int64_t P = t.it[e].P;
        const float beta = t.it[e].beta;
        float s = 0.f;
        if (col < n)
            for (int64_t p = rl; p < P; p += 16) s += partial[p * n + col];
        __syncthreads();                 // the previous chain member's sums have been read
        red[rl][cl] = s;
        __syncthreads();
        if (rl == 0 && col < n) {
            float tot = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) tot += red[k][cl];
            const float prev = first ? (beta != 0.f ? out[ocol] : 0.f) : carried;
            carried = (beta != 0.f ? beta * prev : 0.f) + tot;
        }
        first = false;
        e = t.it[e].next;
    }
    if (rl == 0 && col < n) out[ocol] = carried;
}

void launch_batch(const ReduceItem* items, int count, hipStream_t st) {
    ReduceTable t;
    int64_t nmax = 0;
    for (int i = 0; i < kTable; ++i) {
        t.it[i] = i < count ? items[i] : ReduceItem{};
        t.it[i].next = -1;
        t.it[i].head = i < count ? 1 : 0;
        if (i < count && items[i].n > nmax) nmax = items[i].n;
    }
    // link the chains (same output), in recording order
    for (int i = 0; i < count; ++i) {
        if (!t.it[i].head) continue;
        int last = i;
        for (int j = i + 1; j < count; ++j)
            if (t.it[j].head && t.it[j].out == t.it[i].out) {
                if (t.it[j].n != t.it[i].n || t.it[j].taps_c != t.it[i].taps_c) {       // never seen: one output reduced at two widths — run the batch item by item
                    for (int k = 0; k < count; ++k) {
                        if (items[k].taps_c > 0) launch_reduce_partials_taps(items[k].partial, items[k].out, items[k].P, items[k].n, items[k].beta, items[k].taps_c, st);
                        else launch_reduce_partials(items[k].partial, items[k].out, items[k].P, items[k].n, items[k].beta, st);
                    }
                    return;
                }
                t.it[last].next = j;
                t.it[j].head = 0;
                last = j;
            }
    }
    hipLaunchKernelGGL(reduce_partials_batched_kernel, dim3((unsigned)cdiv(nmax, 64), (unsigned)count), dim3(1024), 0, st, t);
}
}  // namespace

float* partials_alloc(void* workspace, int64_t bytes) {
    DeferState& d = g_defer;
    if (!d.active || bytes <= 0) return (float*)workspace;
    const int64_t need = (bytes + 255) & ~(int64_t)255;
    if (d.off + need > d.cap || d.n >= kMaxItems - 2) return (float*)workspace;     // no room: this reduction runs at once
    float* p = (float*)(d.arena + d.off);
    d.off += need;
    return p;
}

static bool in_arena(const void* p) {
    const DeferState& d = g_defer;
    return d.active && (const char*)p >= d.arena && (const char*)p < d.arena + d.cap;
}

// run what has been recorded so far (recording order is execution order: a reduction that cannot be deferred must not overtake them).
// The batch runs on the stream the items were RECORDED on — their partial sums were produced there — whatever stream the caller is on.
static void flush_recorded(hipStream_t) {
    DeferState& d = g_defer;
    for (int i = 0; i < d.n; i += kTable) launch_batch(d.items + i, d.n - i < kTable ? d.n - i : kTable, d.stream);
    d.n = 0;
    d.have_stream = false;
}

// true when an item produced on `st` may join the recorded ones
static bool same_stream(hipStream_t st) {
    DeferState& d = g_defer;
    if (!d.have_stream) { d.stream = st; d.have_stream = true; return true; }
    return d.stream == st;
}

void ordered_before_launch(hipStream_t st) {
    if (g_defer.active && g_defer.n > 0) flush_recorded(st);
}

void reduce_or_defer(const float* partial, float* out, int64_t P, int64_t n, float beta, hipStream_t st) {
    if (!in_arena(partial) || g_defer.n >= kMaxItems || P >= (1ll << 31) || n >= (1ll << 31) || !same_stream(st)) {
        if (g_defer.active && g_defer.n > 0) flush_recorded(st);
        launch_reduce_partials(partial, out, P, n, beta, st);
        return;
    }
    ReduceItem it{};
    it.partial = partial; it.out = out; it.P = (int32_t)P; it.n = (int32_t)n; it.beta = beta; it.next = -1; it.head = 1;
    g_defer.items[g_defer.n++] = it;
}

void reduce_taps_or_defer(const float* partial, float* out, int64_t P, int64_t n, float beta, int C, hipStream_t st) {
    if (!in_arena(partial) || g_defer.n >= kMaxItems || P >= (1ll << 31) || n >= (1ll << 31) || !same_stream(st)) {
        if (g_defer.active && g_defer.n > 0) flush_recorded(st);
        launch_reduce_partials_taps(partial, out, P, n, beta, C, st);
        return;
    }
    ReduceItem it{};
    it.partial = partial; it.out = out; it.P = (int32_t)P; it.n = (int32_t)n; it.beta = beta; it.next = -1; it.head = 1; it.taps_c = C;
    g_defer.items[g_defer.n++] = it;
}

void reduce_pair_or_defer(const float* p0, float* out0, const float* p1, float* out1, int64_t P, int64_t n, float beta, hipStream_t st) {
    if (!in_arena(p0) || !in_arena(p1) || g_defer.n + 2 > kMaxItems || !same_stream(st)) {
        if (g_defer.active && g_defer.n > 0) flush_recorded(st);
        launch_reduce_partials_pair(p0, out0, p1, out1, P, n, beta, st);
        return;
    }
    reduce_or_defer(p0, out0, P, n, beta, st);
    reduce_or_defer(p1, out1, P, n, beta, st);
}
}  // namespace dyn

extern "C" int dyn_reduce_defer_begin(void* arena, int64_t arena_bytes) {
    DYN_REQUIRE(arena && arena_bytes >= 256 && (((uintptr_t)arena) & 255) == 0, DYN_E_ARG, "dyn_reduce_defer_begin: need a 256-byte aligned arena");
    DYN_REQUIRE(!dyn::g_defer.active, DYN_E_ARG, "dyn_reduce_defer_begin: a deferral context is already open on this thread");
    dyn::g_defer.active = true;
    dyn::g_defer.arena = (char*)arena;
    dyn::g_defer.cap = arena_bytes;
    dyn::g_defer.off = 0;
    dyn::g_defer.n = 0;
    dyn::g_defer.have_stream = false;
    return DYN_OK;
}

extern "C" int dyn_reduce_defer_flush(void* stream) {
    dyn::DeferState& d = dyn::g_defer;
    if (!d.active) return DYN_OK;
    d.active = false;
    if (d.n > 0 && d.have_stream && d.stream != (hipStream_t)stream) {
        // the partial sums were produced on another stream: run the batch there (ordered after its producers) and say so — the caller's
        // stream has no dependency on the outputs
        dyn::flush_recorded(d.stream);
        dyn::set_error("dyn_reduce_defer_flush: %s", "items were recorded on a different stream than the flush; the batch ran on the recording stream");
        return DYN_E_ARG;
    }
    // batches of 96 in recording order; a chain cut by a batch boundary continues in the next launch, which the stream runs after this one
    dyn::flush_recorded((hipStream_t)stream);
    d.n = 0;
    return dyn::check_launch("dyn_reduce_defer_flush");
}

extern "C" int dyn_reduce_defer_abort(void) {
    dyn::g_defer.active = false;
    dyn::g_defer.have_stream = false;
    dyn::g_defer.n = 0;
    return DYN_OK;
}
