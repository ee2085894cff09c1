/*
 * dyneval.h — C-ABI of libdyneval_hip.so (MI355X / gfx950, hand-written HIP).
 *
 * This is the drop-in boundary UNDER the reference's Python call convention
 *   eval_fn(args, model, spec, seq_len, overlap, tokenizer, ...)      (reference lcasr/lib.py:450-462,640)
 * The reference has no FFI of its own; each entry point below names the reference line(s)
 * whose arithmetic it replaces.  Conventions shared by every entry:
 *   - extern "C", plain pointers + int64 sizes + float scalars, no torch types;
 *   - every pointer is DEVICE memory owned by the caller (PyTorch is only the allocator), row-major, fp32
 *     unless the name says otherwise; the library never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - return 0 (DYN_OK) or a negative DYN_E_* code; dyn_last_error() gives a thread-local message;
 *   - workspaces are caller-provided; their size comes from the matching *_workspace_bytes() query.
 */
#ifndef DYNEVAL_H_
#define DYNEVAL_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYN_OK 0
#define DYN_E_ARG (-1)       /* bad argument (shape, alignment, null pointer) */
#define DYN_E_LAUNCH (-2)    /* hipLaunch reported an error */
#define DYN_E_WORKSPACE (-3) /* workspace too small */
#define DYN_E_UNSUPPORTED (-4)

const char* dyn_last_error(void);
const char* dyn_version(void);
const char* dyn_arch(void); /* "gfx950" */

/* ------------------------------------------------------------------------------------------------
 * Dense fp32 GEMM on MFMA (v_mfma_f32_32x32x2_f32): the encoder's QKV / FFN / pointwise-conv / CTC-head
 * products and their dgrad / wgrad — replaces the torch.nn.Linear / Conv1d(k=1) calls inside
 * model(audio_signal=...) (reference lcasr/lib.py:550,603) and inside loss.backward() (lib.py:579).
 *
 *   C[z] = alpha * opA(A[z]) @ opB(B[z]) + beta * C[z] (+ bias[n])       z = z1 * nb2 + z2
 * opA(A) is M x K: trans_a == 0 -> A stored [M, K] (row stride lda); trans_a == 1 -> A stored [K, M].
 * opB(B) is K x N: trans_b == 0 -> B stored [K, N] (row stride ldb); trans_b == 1 -> B stored [N, K].
 * Two-level batch: operand offset = z1 * s?1 + z2 * s?2 (elements) — lets attention heads be addressed
 * inside a [B, T, H*D] activation without a transpose copy.
 * split_k > 1 reduces K-slices through `workspace` in a fixed order (deterministic, no atomics).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t trans_a, trans_b;
    int64_t M, N, K;
    float alpha, beta;
    const float* A; int64_t lda, sa1, sa2;
    const float* B; int64_t ldb, sb1, sb2;
    float* C;       int64_t ldc, sc1, sc2;
    const float* bias;      /* [N] or NULL; added once (after alpha/beta) */
    int64_t nb1, nb2;       /* batch = nb1 * nb2 (>= 1 each) */
    int32_t split_k;        /* 0/1 = none */
    void* workspace; int64_t workspace_bytes;
} dyn_gemm_desc;

int64_t dyn_gemm_f32_workspace_bytes(const dyn_gemm_desc* d);
int dyn_gemm_f32(const dyn_gemm_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNEVAL_H_ */
