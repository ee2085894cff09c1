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
    int32_t split_k;        /* 0 = let the planner choose, n >= 1 = force n K-slices over the whole problem */
    void* workspace; int64_t workspace_bytes;
    /* optional forced configuration (0 = planner: tuned table for known shapes, cost model otherwise); used by
     * scripts/tune_gemm.py to time candidates: tile 64|128 x 64|128, tail_slices = K-slices of the last partial
     * round of workgroups (1 = off) */
    int32_t tile_m, tile_n, tail_slices, reserved_;
    /* optional residual source: C = alpha*A@B + beta*C_in (+ bias), C_in addressed exactly like C (same ldc / batch
     * strides); NULL = accumulate in place.  Lets `x_new = x + f(x)` keep x intact for the backward without a copy. */
    const float* C_in;
    /* activation fused into the store of C (v = alpha*A@B + beta*C_in + bias):
     *   0 none;  1 C = silu(v) and, if aux != NULL, aux = v (the pre-activation the backward needs);  2 C = v * silu'(aux)
     * aux is addressed exactly like C.  Replaces the separate SiLU kernels around the FFN products (the `F.silu` between the two
     * FFN linears inside model(audio_signal=...) and its autograd backward, reference lcasr/lib.py:550,579). */
    int32_t epilogue, reserved2_;
    float* aux;
    /* grouped launches only (dyn_gemm_f32_grouped, trans_a): a_colsum[m] = a_colsum_beta * a_colsum[m] + sum_k A(k, m), i.e. the bias
     * gradient of the linear layer whose weight gradient this product is, summed from the A panel the product streams anyway. */
    float* a_colsum;
    float a_colsum_beta, reserved3_;
    /* optional: n_counters int32 arrival counters owned by the caller, ZERO before the first call and left zero by every call
     * (do not share them between streams).  With them the K slices of a split / tail-sliced product are combined inside the
     * kernel by the workgroup that arrives last (slices summed in slice order: same bits as the separate reduce pass, which is
     * used when counters == NULL or there are fewer counters than partial tiles). */
    int32_t* counters;
    int64_t n_counters;
    /* batch strides of `bias` in elements (0, 0 = the same [N] row for every batch): a product batched over SEVERAL WEIGHT MATRICES —
     * the lockstep group of recordings, each with its own adapted weights (sb1 / sb2 step through the weights) — adds each weight's own bias. */
    int64_t bias_s1, bias_s2;
} dyn_gemm_desc;

int64_t dyn_gemm_f32_workspace_bytes(const dyn_gemm_desc* d);
int dyn_gemm_f32(const dyn_gemm_desc* d, void* stream);
/* n <= 96 independent products (same trans flags, no batching / split-K / bias / activation) as ONE launch over all their 128x128
 * tiles: the weight-gradient products of a whole backward pass (`loss.backward()`, reference lcasr/lib.py:579), each of which alone
 * has too few tiles to fill 256 CUs.  `descs` is a HOST array; the descriptor table is rebuilt in `workspace` by a kernel taking it
 * as arguments (hipGraph-capturable).  Outputs must not alias each other.  Deterministic (no atomics, fixed summation order). */
int64_t dyn_gemm_f32_grouped_workspace_bytes(int32_t n);
int dyn_gemm_f32_grouped(const dyn_gemm_desc* descs, int32_t n, void* workspace, int64_t workspace_bytes, void* stream);

/* EXPERIMENTAL (not on the product path): C[M, N] = X[M, K] . W[N, K]^T (+ bias[N]) as an fp32-grade product on the bf16
 * matrix cores — every fp32 operand split into three bf16 terms, six v_mfma_f32_32x32x16_bf16 per fp32 product, fp32 accumulation
 * (the nn.Linear products inside model(audio_signal=...), reference lcasr/lib.py:550).  K % 32 == 0, X / W 16-byte aligned, ldx / ldw
 * multiples of 4, finite operands.  First, unpipelined form (end of round 4): tests/test_zz_experimental_bf16x3_gpu.py holds it to float64 next to
 * dyn_gemm_f32; scripts/emulate_bf16x3_kernel.py is the lane-level CPU emulation of its indexing. */
int dyn_gemm_bf16x3_nt(const float* X, const float* W, const float* bias, float* C, int64_t M, int64_t N, int64_t K, int64_t ldx,
                       int64_t ldw, int64_t ldc, void* stream);
/* The same for C[M, N] = op(A) . op(B) with dyn_gemm_f32's transpose flags (trans_a: A stored [K][M]; trans_b: B stored [N][K]): the linear layer's
 * input gradient is (0, 0), its weight gradient (1, 0) (loss.backward(), reference lcasr/lib.py:579).  Only (0, 1) has run on hardware so far. */
int dyn_gemm_bf16x3(int trans_a, int trans_b, const float* A, const float* B, const float* bias, float* C, int64_t M, int64_t N, int64_t K,
                    int64_t lda, int64_t ldb, int64_t ldc, void* stream);
/* EXPERIMENTAL, not yet run on hardware: the weight operand split once (planes = unsigned short [3][rows][K], K contiguous: the three bf16 terms of
 * src [rows][K]) instead of by every workgroup that stages it, and X W^T (+ bias) on those planes (rows = N).  The weights change once per
 * optimiser step (reference lcasr/lib.py:580), so the split is one HBM pass per step. */
int dyn_bf16x3_split(const float* src, void* planes, int64_t rows, int64_t K, int64_t ld, void* stream);
int dyn_gemm_bf16x3_presplit(const float* X, const void* w_planes, const float* bias, float* C, int64_t M, int64_t N, int64_t K, int64_t ldx,
                             int64_t ldc, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HBM-bound encoder pieces (activations, norms, softmax, convolutions).  All replace ops inside
 * model(audio_signal=...) (reference lcasr/lib.py:550,603) and its autograd backward (lib.py:579);
 * the architecture constants come from earnings_finetune/lcasr160rb1.yaml:1-29.
 * "beta" arguments accumulate into the destination (dst = result + beta * dst), which is how residual
 * gradients and repeated weight-gradient contributions are summed without extra passes.
 * ------------------------------------------------------------------------------------------------ */
int dyn_silu_fwd(const float* x, float* y, int64_t n, void* stream);
int dyn_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* GLU over the last dim: u [rows, 2C] -> y [rows, C] = u[:, :C] * sigmoid(u[:, C:]) */
int dyn_glu_fwd(const float* u, float* y, int64_t rows, int64_t C, void* stream);
int dyn_glu_bwd(const float* u, const float* dy, float* du, int64_t rows, int64_t C, void* stream);
/* y = a * x + b * y */
int dyn_axpby(const float* x, float* y, float a, float b, int64_t n, void* stream);
/* out[C] = beta * out + sum_rows x[rows, C]   (bias gradients; deterministic two-stage) */
int64_t dyn_colsum_workspace_bytes(int64_t rows, int64_t C);
int dyn_colsum(const float* x, float* out, int64_t rows, int64_t C, float beta, void* workspace, int64_t workspace_bytes,
               void* stream);
int dyn_reduce_partials(const float* partial, float* out, int64_t P, int64_t n, float beta, void* stream);
/* Deferred column reductions.  The reductions that end dyn_layernorm_bwd / dyn_layernorm_bwd_res / dyn_rmsnorm_bwd / dyn_chanaffine_bwd (weight gradients) and
 * dyn_colsum (bias gradients) / dyn_dwconv1d_wgrad, dyn_dwconv2d_s2_wgrad, dyn_sub12_bwd (convolution taps) are launch-bound (~7 us for a few hundred KB, ~55 per window of the adapt step) and nothing reads their
 * outputs before the optimiser step (`loss.backward()` ... `optimizer.step()`, reference lcasr/lib.py:579-581).  Between begin and flush
 * (thread-local, not nestable) those entry points keep their partial sums in `arena` (256-byte aligned, untouched by anything else until
 * the flush) and only RECORD the reduction; flush runs the recorded ones as one launch per 96, each with the per-column summation order
 * of the separate launch, reductions into the same output chained in recording order: bit-identical results.  When the arena is full the
 * entry points fall back to reducing at once.  abort drops the context without launching (error paths). */
int dyn_reduce_defer_begin(void* arena, int64_t arena_bytes);
int dyn_reduce_defer_flush(void* stream);
int dyn_reduce_defer_abort(void);
/* [F, T] (row stride ldx) -> [T, F]: a log-mel window enters the encoder channels-last */
int dyn_transpose_ft(const float* x, float* y, int64_t F, int64_t T, int64_t ldx, void* stream);

/* LayerNorm / RMSNorm over C (C % 256 == 0, C <= 2048); mean/rstd [rows] are saved for the backward. */
int64_t dyn_norm_bwd_workspace_bytes(int64_t rows, int64_t C);
int dyn_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                      int64_t rows, int64_t C, float eps, void* stream);
int dyn_layernorm_bwd(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy, float* dx,
                      float dx_beta, float* dgamma, float* dbeta, float wgrad_beta, int64_t rows, int64_t C,
                      void* workspace, int64_t workspace_bytes, void* stream);
/* same, out of place: dx = norm_bwd(dy) + dx_beta * dx_in.  The residual-stream gradient of a block gets a NEW buffer per module,
 * so the previous one stays intact as the A operand of the weight-gradient products deferred to the end of the backward
 * (dyn_gemm_f32_grouped). */
int dyn_layernorm_bwd_res(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy,
                          const float* dx_in, float* dx, float dx_beta, float* dgamma, float* dbeta, float wgrad_beta,
                          int64_t rows, int64_t C, void* workspace, int64_t workspace_bytes, void* stream);
int dyn_rmsnorm_fwd(const float* x, const float* gamma, float* y, float* rstd, int64_t rows, int64_t C, float eps,
                    void* stream);
int dyn_rmsnorm_bwd(const float* x, const float* gamma, const float* rstd, const float* dy, float* dx, float dx_beta,
                    float* dgamma, float wgrad_beta, int64_t rows, int64_t C, void* workspace, int64_t workspace_bytes,
                    void* stream);
/* Lockstep-group variants (r04).  Several recordings advance through the same window step in ONE batch, each with its own adapted weights
 * (recordings are independent: fresh optimiser and restored weights per eval_fn call, reference lcasr/lib.py:494,636-637).  `rows` =
 * n_samples * rows_per_sample; sample s takes the parameters of group s % n_groups, the groups' parameters lie param_stride elements apart
 * (the group's flat parameter buffer is [n_groups, n_flat]); weight gradients of group g go to dgamma + g * param_stride (a group's second
 * sample accumulates onto its first).  n_groups = 1 is exactly the plain entry; per sample the arithmetic, the partial-sum layout and the
 * reduction order are those of a plain launch over that sample's rows: bit-identical to running the recordings one by one. */
int64_t dyn_norm_bwd_workspace_bytes_g(int64_t rows, int64_t C, int64_t rows_per_sample);
int dyn_layernorm_fwd_g(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows, int64_t C,
                        float eps, int64_t rows_per_sample, int64_t n_groups, int64_t param_stride, void* stream);
int dyn_layernorm_bwd_g(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy, const float* dx_in,
                        float* dx, float dx_beta, float* dgamma, float* dbeta, float wgrad_beta, int64_t rows, int64_t C,
                        int64_t rows_per_sample, int64_t n_groups, int64_t param_stride, void* workspace, int64_t workspace_bytes, void* stream);
int dyn_rmsnorm_bwd_g(const float* x, const float* gamma, const float* rstd, const float* dy, float* dx, float dx_beta, float* dgamma,
                      float wgrad_beta, int64_t rows, int64_t C, int64_t rows_per_sample, int64_t n_groups, int64_t param_stride,
                      void* workspace, int64_t workspace_bytes, void* stream);
/* BatchRenorm1d in eval mode = per-channel affine with the running statistics (the loop calls model.eval(), reference
 * lcasr/lib.py:525, so they are constants):  y = (x - mean_c) * rsqrt(var_c + eps) * weight_c + bias_c;  x, y [rows, C].
 * bwd: dx = dy * rsqrt(var + eps) * weight (+ dx_beta * dx), dweight += sum_r dy * xhat, dbias += sum_r dy (deterministic). */
int dyn_chanaffine_fwd(const float* x, const float* mean, const float* var, const float* weight, const float* bias, float* y,
                       int64_t rows, int64_t C, float eps, void* stream);
int64_t dyn_chanaffine_bwd_workspace_bytes(int64_t rows, int64_t C);
int dyn_chanaffine_bwd(const float* x, const float* mean, const float* var, const float* weight, const float* dy, float* dx,
                       float dx_beta, float* dweight, float* dbias, float wgrad_beta, int64_t rows, int64_t C, float eps,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* Row softmax / log-softmax (row length L <= 16384 fwd, <= 8192 bwd).  softmax_bwd: dx = y*(dy - sum(dy*y))*scale;
 * log_softmax_bwd: dx = dy - exp(y)*sum(dy).  F.log_softmax call sites: reference wav2vec2/lib.py:169,417. */
int dyn_softmax_fwd(const float* x, float* y, int64_t rows, int64_t L, int64_t ldx, int64_t ldy, void* stream);
int dyn_softmax_bwd(const float* y, const float* dy, float* dx, int64_t rows, int64_t L, int64_t ld, float scale,
                    void* stream);
/* Key-length masked softmax (the attention of a zero-padded utterance bucket, wav2vec2_model.py; HF passes no attention mask to
 * wav2vec2-base, reference wav2vec2/lib.py:413, so an utterance attends to exactly its own frames): `valid_cols` is a DEVICE int32
 * scalar read when the kernel runs; columns >= *valid_cols are outside the max / sum and get probability 0, the first *valid_cols
 * values are bit for bit those of a row of that length. */
int dyn_softmax_fwd_len(const float* x, float* y, int64_t rows, int64_t L, int64_t ldx, int64_t ldy, const int32_t* valid_cols,
                        void* stream);
int dyn_log_softmax_fwd(const float* x, float* y, int64_t rows, int64_t L, int64_t ldx, int64_t ldy, void* stream);
int dyn_log_softmax_bwd(const float* y, const float* dy, float* dx, int64_t rows, int64_t L, int64_t ld, void* stream);
/* Gradient of the mean categorical entropy w.r.t. the log-probabilities (the `entropy_augmentation` input perturbation,
 * reference lcasr/lib.py:86-99): grad[r, c] = -p (y + H_r) * scale, H_r = -sum p y; entropy_per_row optional. */
int dyn_entropy_grad(const float* log_probs, float* grad, float* entropy_per_row, int64_t rows, int64_t L, int64_t ld,
                     float scale, void* stream);

/* Depthwise Conv1d over time, channels-last x [B, T, C], w [C, KW], 'same' zero padding (conformer conv module,
 * `conv_kernel_size: 9`, yaml:15).  KW in {3,5,7,9,15,31}. */
int dyn_dwconv1d_fwd(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t T, int64_t C,
                     int64_t KW, void* stream);
int dyn_dwconv1d_dgrad(const float* dy, const float* w, float* dx, int64_t B, int64_t T, int64_t C, int64_t KW,
                       float dx_beta, void* stream);
int64_t dyn_dwconv1d_wgrad_workspace_bytes(int64_t B, int64_t T, int64_t C, int64_t KW);
int dyn_dwconv1d_wgrad(const float* x, const float* dy, float* dw, float* dbias, float beta, int64_t B, int64_t T, int64_t C,
                       int64_t KW, void* workspace, int64_t workspace_bytes, void* stream);
/* lockstep-group variants (see dyn_layernorm_fwd_g): sample b convolves with / accumulates into the filters of replica b % n_groups */
int dyn_dwconv1d_dgrad_g(const float* dy, const float* w, float* dx, int64_t B, int64_t T, int64_t C, int64_t KW, float dx_beta,
                         int64_t n_groups, int64_t param_stride, void* stream);
int dyn_dwconv1d_wgrad_g(const float* x, const float* dy, float* dw, float* dbias, float beta, int64_t B, int64_t T, int64_t C, int64_t KW,
                         int64_t n_groups, int64_t param_stride, void* workspace, int64_t workspace_bytes, void* stream);

/* Fused conformer conv-module core (GLU -> depthwise k=9 -> RMSNorm | LayerNorm over channels -> SiLU), one pass:
 *   u [B, T, 2C] -> s [B, T, C];  optional outputs for the backward: g (GLU), c (conv), nn (norm output), mean, rstd [B*T].
 * C in {256, 512, 768, 1024}, kernel width 9. */
int dyn_convmod_fwd(const float* u, const float* w, const float* bias, const float* gamma, const float* beta, float* s,
                    float* g_out, float* c_out, float* nn_out, float* mean_out, float* rstd_out, int64_t B, int64_t T, int64_t C,
                    int64_t KWIDTH, int32_t layernorm, float eps, void* stream);
/* lockstep-group variant (see dyn_layernorm_fwd_g): sample b of the batch takes w / bias / gamma / beta of replica b % n_groups */
int dyn_convmod_fwd_g(const float* u, const float* w, const float* bias, const float* gamma, const float* beta, float* s, float* g_out,
                      float* c_out, float* nn_out, float* mean_out, float* rstd_out, int64_t B, int64_t T, int64_t C, int64_t KWIDTH,
                      int32_t layernorm, float eps, int64_t n_groups, int64_t param_stride, void* stream);

/* dw_striding x8 subsampling (`subsampling: dw_striding`, `subsampling_conv_channels: 256`, `subsampling_act: silu`,
 * yaml:10-13), channels-last.  Output sizes: To = (T-1)/2+1, Fo = (F-1)/2+1 (3x3, stride 2, pad 1).
 *   conv2d_first: x [B,T,F] (1 channel) -> z [B,To,Fo,C], w [C,3,3]
 *   dwconv2d_s2 : u = bias + dw3x3_s2(silu(z)),  z [B,T,F,C] -> u [B,To,Fo,C]   (SiLU fused into the load)
 * The 1x1 pointwise convs between them are dyn_gemm_f32 calls. */
int dyn_conv2d_first_fwd(const float* x, const float* w, const float* bias, float* z, int64_t B, int64_t T, int64_t F,
                         int64_t C, void* stream);
/* input gradient of conv2d_first (needed only by the entropy-gradient input perturbation, reference lib.py:96) */
int dyn_conv2d_first_dgrad(const float* dz, const float* w, float* dx, int64_t B, int64_t T, int64_t F, int64_t C,
                           void* stream);
int64_t dyn_conv2d_wgrad_workspace_bytes(int64_t B, int64_t To, int64_t C);
int dyn_conv2d_first_wgrad(const float* x, const float* dz, float* dw, float* dbias, float beta, int64_t B, int64_t T,
                           int64_t F, int64_t C, void* workspace, int64_t workspace_bytes, void* stream);
/* The first two subsampling stages FUSED: u2 = bias2 + dw3x3_s2(silu(bias1 + conv3x3_s2(x))), x [B, T, F] one channel ->
 * u2 [B, T2, F2, C] (T1 = (T - 1) / 2 + 1, T2 = (T1 - 1) / 2 + 1, same for F).  The [B, T1, F1, C] intermediate (671 MB at B = 2,
 * T = 16384: the largest activation of the model) and its gradient never exist: forward and backward recompute it from x
 * (reference: upstream SCConformerXL `subsampling` reached through model(audio_signal=...) and loss.backward(), lcasr/lib.py:550,579).
 * dyn_sub12_bwd accumulates dw1 [C, 3, 3], db1 [C], dw2 [C, 3, 3], db2 [C] (beta * old + new) from du2; deterministic. */
int dyn_sub12_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* u2, int64_t B, int64_t T,
                  int64_t F, int64_t C, void* stream);
int64_t dyn_sub12_bwd_workspace_bytes(int64_t B, int64_t T, int64_t C);
int dyn_sub12_bwd(const float* x, const float* du2, const float* w1, const float* b1, const float* w2, float* dw1, float* db1, float* dw2,
                  float* db2, float beta, int64_t B, int64_t T, int64_t F, int64_t C, void* workspace, int64_t workspace_bytes, void* stream);
int dyn_dwconv2d_s2_fwd(const float* z, const float* w, const float* bias, float* u, int64_t B, int64_t T, int64_t F,
                        int64_t C, void* stream);
int dyn_dwconv2d_s2_dgrad(const float* z, const float* w, const float* du, float* dz, int64_t B, int64_t T, int64_t F,
                          int64_t C, void* stream);
int dyn_dwconv2d_s2_wgrad(const float* z, const float* du, float* dw, float* dbias, float beta, int64_t B, int64_t T,
                          int64_t F, int64_t C, void* workspace, int64_t workspace_bytes, void* stream);

/* Rotary position embedding, in place on the first n_heads*D floats of each row of x ([B*T] rows of row_stride floats);
 * cos/sin tables are [T, D/2]; inverse != 0 applies the transpose (the backward). yaml:22,28. */
int dyn_rotary(float* x, const float* cos_table, const float* sin_table, int64_t B, int64_t T, int64_t n_heads, int64_t D,
               int64_t row_stride, int32_t inverse, void* stream);

/* SpecAugment frequency masks on a [F, T] log-mel window, in place: rows f0[k] <= f < f0[k]+width[k] := value.
 * Replaces lcasr.utils.augmentation.SpecAugment as called at reference lcasr/lib.py:499,541 (mask positions are
 * drawn on the host so the RNG stream stays the caller's).  value_dev (nullable) overrides `value` with a DEVICE scalar,
 * e.g. the window mean produced by dyn_moments, so the fill value never makes a host round trip. */
int dyn_specaug_freqmask(float* x, int64_t F, int64_t T, const int32_t* f0, const int32_t* width, int64_t n_masks,
                         float value, const float* value_dev, void* stream);
int dyn_specaug_timemask(float* x, int64_t F, int64_t T, const int32_t* t0, const int32_t* width, int64_t n_masks,
                         float value, const float* value_dev, void* stream);

/* Same masking with the (start, width) pairs passed by value as kernel arguments: start_host / width_host are HOST arrays
 * (n_masks <= 32) read at launch time, so the per-window loop has no device index buffer and no blocking
 * host-to-device copy (a pageable copy would make the host wait for the previous window's backward). */
int dyn_specaug_mask_args(float* x, int64_t F, int64_t T, const int32_t* start_host, const int32_t* width_host, int64_t n_masks,
                          int32_t along_time, float value, const float* value_dev, void* stream);

/* Log-mel front end (upstream lcasr.utils.audio_tools.processing_chain, called by the dataset adapters: reference
 * lcasr/earnings22/run.py:61, tedlium/run.py:94, chime6/run.py:61-68).  The STFT itself is a dyn_gemm_f32 call on the
 * reflect-padded signal with lda = hop < K (overlapping rows) and a window-folded DFT basis; these are the passes
 * around it:  reflect pad -> [GEMM: re|im] -> power -> [GEMM: mel] -> log + per-bin (mean, std) -> normalise + transpose.
 *   dyn_stft_power     reim [T, 2*KP] (re | im) -> power [T, KP]
 *   dyn_logmel_finish  mel [T, F] (overwritten by log(mel + eps)) -> out [F, T], optionally per-bin normalised over T */
int dyn_reflect_pad(const float* x, float* out, int64_t n, int64_t pad, void* stream);
int dyn_stft_power(const float* reim, float* power, int64_t T, int64_t KP, void* stream);
int64_t dyn_logmel_finish_workspace_bytes(int64_t T, int64_t F);
int dyn_logmel_finish(float* mel, float* out, int64_t T, int64_t F, float eps, int32_t normalize, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* (x - mean_row) / std_row (unbiased) of a [R, T] spectrogram: the renormalisation after the CHiME-6 channel average
 * (reference lcasr/chime6/run.py:66-68).  In place when out == x. */
int dyn_rownorm(const float* x, float* out, int64_t R, int64_t T, void* stream);

/* Optional Loop-A augmentations on a device-resident [F, T] window (random draws stay on the host):
 *   dyn_gather_frames  frame_shuffle (reference lcasr/lib.py:81-84): y = x[:, index] (along_time) or x[index, :]
 *   dyn_moments        (sum, mean, unbiased std) of a buffer — spec.std() of add_random_noise (lib.py:379-382)
 *   dyn_cutout         cutout (lib.py:384-417): rects [n, 4] = (y0, y1, x0, x1); mode 0 zero, 1 own mean, 2 constant */
int dyn_gather_frames(const float* x, const int32_t* index, float* y, int64_t F, int64_t T, int32_t along_time, void* stream);
int64_t dyn_moments_workspace_bytes(void);
int dyn_moments(const float* x, int64_t n, float* out3, void* workspace, int64_t workspace_bytes, void* stream);
int dyn_cutout(float* x, int64_t F, int64_t T, const int32_t* rects, int64_t n_rects, int32_t mode, float value,
               float* means_scratch, void* stream);

/* Device-side delay of `microseconds` on `stream` (one wave polling the 100 MHz realtime counter): starts the recording chains of
 * lib.dynamic_eval_many out of phase.  No reference counterpart (the reference runs one recording at a time,
 * run_dynamic_eval_full.py:84-100); scheduling aid only, computes nothing. */
int dyn_sleep_us(int64_t microseconds, void* stream);
/* A HIP stream restricted to the compute units set in `mask` (bit i of word i / 32 = CU i, hipExtStreamCreateWithCUMask), and its
 * destruction.  lib.dynamic_eval_many can run its recording chains on such streams (DYN_CHAIN_CU_MASK=<hole size>): chain k's kernels
 * then leave a different group of CUs free, on which the short kernels of the other chains can start while a matrix kernel of chain k
 * owns the rest of the chip.  No reference counterpart; scheduling aid only (A/B result in DESIGN.md §5). */
int dyn_stream_create_cu_mask(const uint32_t* mask, int32_t n_words, void** stream_out);
int dyn_stream_destroy(void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused self-attention forward for no-grad passes (reference lcasr/lib.py:603: the final pass runs model(audio_signal) under
 * torch.no_grad(); also every epochs = 0 evaluation): out = softmax(q k^T * scale) v per (batch, head), fp32, online softmax,
 * K/V tiles staged in LDS — the [T, T] scores never reach HBM.  q / k / v are views of [B, T, .] activations with a common row
 * and batch stride (the packed QKV activation: three base pointers), head h at +h * head_dim; head_dim must be 128.
 * ------------------------------------------------------------------------------------------------ */
int dyn_attention_fwd(const float* q, const float* k, const float* v, float* out, int64_t B, int64_t T, int64_t H,
                      int64_t head_dim, int64_t row_stride, int64_t batch_stride, int64_t out_row_stride,
                      int64_t out_batch_stride, float scale, void* stream);
/* Grad-mode variant (the B = 2 forward of every adapt step, reference lcasr/lib.py:550): same kernel, additionally writes
 * lse [B, H, T] = log sum_j exp(scale * q_i k_j) per query row — all the backward needs to re-form the probabilities. */
int dyn_attention_fwd_lse(const float* q, const float* k, const float* v, float* out, float* lse, int64_t B, int64_t T, int64_t H,
                          int64_t head_dim, int64_t row_stride, int64_t batch_stride, int64_t out_row_stride,
                          int64_t out_batch_stride, float scale, void* stream);
/* The same forward with the KEYS split over `nsplit` workgroups per query block (0 = chosen from the launch size, 1 = no split,
 * at most 8): each split writes a normalised partial output and its log-sum-exp to `workspace`, a second kernel merges them in
 * split order (deterministic).  For launches whose (batch, head, query-block) workgroups do not fill the chip: the B = 4 final-pass
 * launch at T' = 2048 (384 workgroups on 512 slots) and every B = 1 inference.  `lse` may be NULL. */
int64_t dyn_attention_fwd_split_workspace_bytes(int64_t B, int64_t T, int64_t H, int32_t nsplit);
int dyn_attention_fwd_split(const float* q, const float* k, const float* v, float* out, float* lse, int64_t B, int64_t T, int64_t H,
                            int64_t head_dim, int64_t row_stride, int64_t batch_stride, int64_t out_row_stride, int64_t out_batch_stride,
                            float scale, int32_t nsplit, void* workspace, int64_t workspace_bytes, void* stream);
/* Backward of the fused attention (`loss.backward()`, reference lcasr/lib.py:579, through softmax(q k^T) v): dq / dk / dv from
 * (q, k, v, out, dout, lse), recomputing P tile by tile from LDS-staged tiles — no [T, T] matrix is read or written.
 * Deterministic (no float atomics): query-owner workgroups produce dq and delta [B, H, T] (scratch: rowsum(dout * out)), then
 * key-owner workgroups produce dk and dv.  dout / out share (out_row_stride, out_batch_stride); dq / dk / dv share
 * (grad_row_stride, grad_batch_stride) — e.g. the three thirds of a packed [B, T, 3 * H * 128] gradient buffer. */
int dyn_attention_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout, const float* lse,
                      float* delta, float* dq, float* dk, float* dv, int64_t B, int64_t T, int64_t H, int64_t head_dim,
                      int64_t row_stride, int64_t batch_stride, int64_t out_row_stride, int64_t out_batch_stride,
                      int64_t grad_row_stride, int64_t grad_batch_stride, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Encoder-decoder `teacher_ce` adaptation (reference lcasr/lib.py:1228-1322 calc_loss_enc_dec, :1475-1732 enc_dec_dynamic_eval):
 * the decoder's dense work runs on dyn_gemm_f32 and the encoder's norm / softmax kernels; these are the pieces that are new.
 *   dyn_embedding_fwd  out[s] = table[ids[s]] (+ pos[s % pos_period])            token embedding (+ fixed positional table)
 *   dyn_embedding_bwd  dtable[v] = beta * dtable[v] + sum_{s: ids[s] == v} dy[s]  one workgroup per vocabulary row, no atomics
 *   dyn_causal_mask    scores[b, r, c] = -inf for c > r                            decoder self-attention (before dyn_softmax_fwd)
 *   dyn_nll_loss       F.cross_entropy(reduction='sum', ignore_index) on log-softmax rows (lib.py:1290-1296): loss [1], the per-row
 *                      terms, and grad = grad_scale * (exp(logp) - onehot(target)) w.r.t. the LOGITS (0 for ignored rows)
 * ------------------------------------------------------------------------------------------------ */
/* ids[r] = argmax_c x[r, c] (first maximum wins, as torch.argmax), vals[r] = that maximum (may be NULL): the greedy step of the
 * autoregressive decode (`model.generate`, reference lcasr/lib.py:1128,1580) and the teacher's mean max-probability (:1601). */
int dyn_argmax_rows(const float* x, int64_t rows, int64_t C, int64_t ld, int32_t* ids, float* vals, void* stream);
int dyn_embedding_fwd(const int32_t* ids, const float* table, const float* pos, float* out, int64_t S, int64_t d, int64_t vocab,
                      int64_t pos_period, void* stream);
int dyn_embedding_bwd(const int32_t* ids, const float* dy, float* dtable, int64_t S, int64_t d, int64_t vocab, float beta,
                      void* stream);
int dyn_causal_mask(float* scores, int64_t nb, int64_t S, void* stream);
int dyn_nll_loss(const float* log_probs, const int32_t* targets, float* loss, float* row_loss, float* grad, int64_t rows, int64_t C,
                 int32_t ignore_index, float grad_scale, void* stream);
/* Counter-based randomness of the teacher_ce path (no device RNG state, a draw is a pure function of (seed, stream, index)):
 *   z = splitmix64_finish(seed ^ (stream + 1) * 0x9E3779B97F4A7C15 ^ (index + 1) * 0xC2B2AE3D27D4EB4F)
 *   dropout:  keep element i iff (z >> 40) * 2^-24 >= p;  y = keep ? x * 1 / (1 - p) : 0   (in place allowed; the backward is the same
 *             call on the gradient).  Replaces nn.Dropout of `language_model_decoder.dropout_emb / ff_out_dropout` and the attention
 *             `dropout_p` the reference switches on around the supervised step (lcasr/lib.py:1511-1522,1636-1637,1703-1707).
 *   sampling: ids[r] = argmax_c (x[r, c] * inv_temperature + g), g = -log(-log(u)), u = (2 * (z >> 41) + 1) * 2^-24 with
 *             stream = step0 + r, index = c  (Gumbel-max = a draw from softmax(x / temperature)): `model.generate(sample=True,
 *             temperature=...)` of the decode-agreement filter (lcasr/lib.py:1620-1627). */
int dyn_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream);
int dyn_gumbel_argmax_rows(const float* x, int64_t rows, int64_t C, int64_t ld, float inv_temperature, uint64_t seed, uint64_t step0,
                           int32_t* ids, void* stream);

/* The autoregressive decode of `model.generate` (reference call sites lcasr/lib.py:1128,1579-1582,1620-1625; the model class itself
 * is un-vendored) one token at a time with cached keys / values: ONE call runs `n_steps` consecutive positions t0 .. t0 + n_steps - 1
 * as 8 * layers + 2 lean launches per token (a row-times-matrix kernel with the LayerNorm / embedding prologue and the bias / SiLU /
 * residual epilogue fused, and a one-query attention kernel per head), instead of ~57 launches of the tile kernels at M = 1.
 * Per position t: x = embed[tokens[t]] + pos_table[t]; per layer: self-attention over cache rows 0 .. t (row t = the packed q | k | v
 * of this position, written here), cross-attention over the n_enc projected encoder rows, SiLU feed-forward; then
 * logits = head(norm_out(x)) and tokens[t + 1] = argmax (sample == 0, first maximum wins) or the Gumbel-max draw of
 * dyn_gumbel_argmax_rows with stream = step0 + t (sample != 0).  Nothing is read back: the caller looks for eos when it likes.
 * layer_ptrs: HOST array of layers * DYN_DEC_PTRS_PER_LAYER device pointers, per layer in this order:
 *   self.norm.weight, self.norm.bias, self.qkv.weight [3d, d], self.qkv.bias, self.out.weight [d, d], self.out.bias,
 *   cross.norm.weight, cross.norm.bias, cross.q.weight [d, d], cross.q.bias, cross.out.weight [d, d], cross.out.bias,
 *   ff.norm.weight, ff.norm.bias, ff.w1.weight [d_ff, d], ff.w2.weight [d, d_ff],
 *   cache [>= t0 + n_steps rows, 3d] (read / written), cross_kv [n_enc, 2d] (keys | values of the encoder states).
 * Limits: d_model % 256 == 0, d_model <= 2048, d_ff % 256 == 0, d_ff <= 2048, head dim a power of two in 4 .. 256,
 * t0 + n_steps <= max_positions, keys per attention <= 49152. */
#define DYN_DEC_PTRS_PER_LAYER 18
typedef struct {
    int32_t d_model, heads, d_ff, vocab, layers, n_enc, max_positions, reserved_;
    float eps, reserved2_;
    const float* embed;        /* [vocab, d] */
    const float* pos_table;    /* [max_positions, d] */
    const float* norm_out_w;
    const float* norm_out_b;
    const float* head_w;       /* [vocab, d] */
    const float* head_b;
    const void* const* layer_ptrs;
    int32_t* tokens;           /* device, >= t0 + n_steps + 1 entries; tokens[t0] must be valid on entry */
    float* logits;             /* device [vocab]: the logits of the last position run */
    float* scratch;            /* device, >= 6 * d_model + d_ff + 16 * heads floats */
    int64_t scratch_floats;
} dyn_decoder_desc;
int dyn_decoder_steps(const dyn_decoder_desc* d, int32_t t0, int32_t n_steps, int32_t sample, float inv_temperature, uint64_t seed,
                      uint64_t step0, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CTC.  dyn_ctc_greedy replaces GreedyCTCDecoder on a CPU copy of the posteriors (reference lcasr/lib.py:498,
 * 559,565; run_dynamic_eval_full.py:53,100): argmax over classes (first maximum), collapse repeats, drop `blank`.
 *   log_probs [B*T rows, C] (row stride ld); argmax_ids [B*T]; out_ids [B, T] (prefix of out_len[b] valid ids).
 * dyn_ctc_loss replaces torch.nn.CTCLoss(blank, reduction) forward + backward (reference lcasr/lib.py:492,575,579;
 * wav2vec2/lib.py:351,434): log_probs[t, b, c] at t*lp_stride_t + b*lp_stride_b + c; targets [B, S_max] int32;
 * reduction 0 = 'sum' (loss = sum_b nll_b, grad scaled by grad_scale), 1 = 'mean' (nll_b / max(S_b,1), mean over B).
 * grad (optional) gets torch's native gradient w.r.t. log_probs, same addressing with g_stride_*.
 * Rounding contract (r04): every exp / log of the lattice and the gradient is glibc's expf / logf bit for bit (csrc/libm_f32.h), every
 * add / subtract one fp32 operation in the order of aten/native/LossCTC.cpp's CPU kernel, the per-class log-sum pairwise in descending
 * lattice position: on identical log_probs the nll, alpha, beta and gradient are bit-identical to torch's CPU CTC.
 * dyn_ctc_loss_workspace_layout: byte offsets of {gathered slab, alpha, beta, nll} inside the workspace and the lattice row length
 * L = 2 * max(S_max, 1) + 1 (alpha / beta are [B, T, L] fp32) — lets a caller or a test read the lattice a dyn_ctc_loss call left behind.
 * dyn_libm_f32: y_exp[i] = expf(x[i]), y_log[i] = logf(x[i]), y_exp_nonpos[i] = the branch-free x <= 0 variant the lattice uses
 * (any output may be NULL): the device build of csrc/libm_f32.h, for comparison with the host's libm.
 * ------------------------------------------------------------------------------------------------ */
int dyn_ctc_greedy(const float* log_probs, int64_t B, int64_t T, int64_t C, int64_t ld, int32_t blank, int32_t* argmax_ids,
                   int32_t* out_ids, int32_t* out_len, void* stream);
int64_t dyn_ctc_loss_workspace_bytes(int64_t T, int64_t B, int64_t S_max);
int dyn_ctc_loss_workspace_layout(int64_t T, int64_t B, int64_t S_max, int64_t* offsets4, int64_t* row_len);
int dyn_libm_f32(const float* x, int64_t n, float* y_exp, float* y_log, float* y_exp_nonpos, void* stream);
int dyn_ctc_loss(const float* log_probs, int64_t T, int64_t B, int64_t C, int64_t lp_stride_t, int64_t lp_stride_b,
                 const int32_t* targets, int64_t S_max, const int32_t* input_lengths, const int32_t* target_lengths,
                 int32_t blank, int32_t reduction, float grad_scale, float* loss, float* nll_per_sample, float* grad,
                 int64_t g_stride_t, int64_t g_stride_b, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused adaptation step over a flat fp32 parameter buffer (`optimizer.step()`, reference lcasr/lib.py:581).
 * `step` is the 0-based count of previous steps; step == 0 initialises the state in-kernel (fresh optimiser per
 * eval_fn call, lib.py:494).  MADGRAD follows facebookresearch/madgrad (lr+eps, cube-root denominator, momentum
 * as an average of the dual-averaged iterate); Adam follows torch.optim.Adam (reference nvidia_ctc/lib.py:43).
 * dyn_clip_grad_norm = torch.nn.utils.clip_grad_norm_ (reference wav2vec2/lib.py:442); norm_and_coef: 2 floats.
 * ------------------------------------------------------------------------------------------------ */
int dyn_madgrad_step(float* params, const float* grads, float* grad_sum, float* grad_sum_sq, float* x0, int64_t n, float lr,
                     float momentum, float weight_decay, float eps, int64_t step, void* stream);
int dyn_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, void* stream);
int64_t dyn_clip_grad_norm_workspace_bytes(int64_t n);
int dyn_clip_grad_norm(float* grads, int64_t n, float max_norm, float* norm_and_coef, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Window stitching on device (reference lcasr/lib.py:583-589,604-609,615-629): acc[pos+r, c] += exp(lp[r, c]),
 * count[pos+r] += 1, then out = log(acc / count) over the covered prefix.
 * ------------------------------------------------------------------------------------------------ */
int dyn_stitch_accumulate(const float* log_probs, int64_t ld, float* acc, float* count, int64_t pos, int64_t rows, int64_t C,
                          int64_t acc_rows, void* stream);
int dyn_stitch_finalize(const float* acc, const float* count, float* out, int64_t rows, int64_t C, void* stream);
/* Same for an accumulator whose coverage has gaps (outer leave-one-out stitching, reference
 * lcasr/run_within_recording_loo_eval.py:160-181): out[r] = log(acc[row_index[r]] / count[row_index[r]]), row_index int64 on device. */
int dyn_stitch_finalize_rows(const float* acc, const float* count, const int64_t* row_index, float* out, int64_t rows, int64_t C,
                             void* stream);

/* ------------------------------------------------------------------------------------------------
 * wav2vec2 encoder pieces (reference wav2vec2/lib.py:20-23 loads HF Wav2Vec2ForCTC; forward at :163,413, backward at
 * :194,437).  The strided Conv1d feature extractor and the grouped positional conv are dyn_gemm_f32 calls over
 * OVERLAPPING rows of channels-last activations (lda = stride * C_in < K = kernel * C_in); these are the passes around
 * them.  Layouts: activations [B, T, C]; conv weights [C_out][kernel][C_in]; positional conv v / w [G * C_out/G][kernel][C/G].
 *   dyn_gelu_*          exact erf GELU
 *   dyn_colnorm_*       GroupNorm(groups == channels): per-(batch, channel) normalisation over time, biased variance
 *   dyn_col2im_1d       input gradient of a strided Conv1d from the dense [B, T_out, kernel*C] gradient of its rows
 *   dyn_group_pack ...  [B, T, C] <-> group-major zero-padded [B, G, T + 2 pad, C/G] (and the gradient counterparts)
 *   dyn_weight_norm_*   w = g[tap] * v / ||v[:, tap, :]||_F  (torch weight_norm with dim = 2) and its backward
 * ------------------------------------------------------------------------------------------------ */
int dyn_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int dyn_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
int64_t dyn_colnorm_workspace_bytes(int64_t B, int64_t T, int64_t C);
int dyn_colnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t B,
                    int64_t T, int64_t C, float eps, void* workspace, int64_t workspace_bytes, void* stream);
int dyn_colnorm_bwd(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy, float* dx,
                    float* dgamma, float* dbeta, float wgrad_beta, int64_t B, int64_t T, int64_t C, void* workspace,
                    int64_t workspace_bytes, void* stream);
/* The same with a DEVICE-side row count (`valid_rows`: null = all T rows, else an int32 scalar in HBM read when the kernels run):
 * statistics / sums over the first *valid_rows rows of every batch entry, dx = 0 past them.  With dyn_mask_rows and
 * dyn_softmax_fwd_len this lets one captured launch sequence (hipGraph) serve every utterance length of a zero-padded bucket of the
 * per-utterance loop (reference wav2vec2/lib.py:293-462 runs each utterance at its own length). */
int dyn_colnorm_fwd_len(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t B,
                        int64_t T, int64_t C, float eps, const int32_t* valid_rows, void* workspace, int64_t workspace_bytes,
                        void* stream);
int dyn_colnorm_bwd_len(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy, float* dx,
                        float* dgamma, float* dbeta, float wgrad_beta, int64_t B, int64_t T, int64_t C, const int32_t* valid_rows,
                        void* workspace, int64_t workspace_bytes, void* stream);
/* x [B, T, C]: rows t >= *valid_rows of every batch entry := 0 (what the positional conv's zero padding holds past the last frame). */
int dyn_mask_rows(float* x, int64_t B, int64_t T, int64_t C, const int32_t* valid_rows, void* stream);
int dyn_col2im_1d(const float* dA, float* dx, int64_t B, int64_t Tin, int64_t Tout, int64_t C, int64_t kw, int64_t stride,
                  void* stream);
int dyn_group_pack(const float* x, float* xg, int64_t B, int64_t T, int64_t C, int64_t G, int64_t pad, void* stream);
int dyn_group_unpack(const float* yg, float* y, const float* bias, int64_t B, int64_t T, int64_t Tg, int64_t C, int64_t G,
                     void* stream);
int dyn_group_pack_grad(const float* dy, float* dyg, int64_t B, int64_t T, int64_t Tg, int64_t C, int64_t G, void* stream);
int dyn_group_unpack_grad(const float* dxg, float* dx, int64_t B, int64_t T, int64_t C, int64_t G, int64_t pad, float beta,
                          void* stream);
int64_t dyn_weight_norm_workspace_bytes(int64_t rows, int64_t kw);
int dyn_weight_norm_fwd(const float* v, const float* g, float* w, int64_t rows, int64_t kw, int64_t cg, void* workspace,
                        int64_t workspace_bytes, void* stream);
int dyn_weight_norm_bwd(const float* v, const float* g, const float* dw, float* dv, float* dg, float beta, int64_t rows,
                        int64_t kw, int64_t cg, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Soft-DTW — replaces the reference's Numba CUDA kernels compute_softdtw_cuda / compute_softdtw_backward_cuda
 * (reference wav2vec2/soft_dtw_cuda.py:33-111), their autograd wrapper (:114-175) and _euclidean_dist_func (:319-329).
 *   dyn_sqdist       D[b,i,j] = sum_k (x[b,i,k] - y[b,j,k])^2            x [B,N,d], y [B,M,d] -> D [B,N,M]
 *   dyn_sqdist_bwd_x dx[b,i,k] = sum_j 2 G[b,i,j] (x[b,i,k] - y[b,j,k])
 *   dyn_softdtw_fwd  R [B,N+2,M+2] fp64 lattice workspace (written entirely by the kernel; keep it for the backward),
 *                    value[b] = R[b,N,M] (fp32)
 *   dyn_softdtw_bwd  E [B,N,M] = d value / d D  (multiply by grad_output on the caller side, as :173-174)
 * bandwidth <= 0 disables the Sakoe-Chiba pruning.  No 1024 limit on N, M (reference :312-314 falls back to CPU).
 * ------------------------------------------------------------------------------------------------ */
int dyn_sqdist(const float* x, const float* y, float* D, int64_t B, int64_t N, int64_t M, int64_t d, void* stream);
int dyn_sqdist_bwd_x(const float* x, const float* y, const float* G, float* dx, int64_t B, int64_t N, int64_t M, int64_t d,
                     void* stream);
int dyn_softdtw_fwd(const float* D, double* R, float* value, int64_t B, int64_t N, int64_t M, float gamma, float bandwidth,
                    void* stream);
int dyn_softdtw_bwd(const float* D, const double* R, float* E, int64_t B, int64_t N, int64_t M, float gamma, float bandwidth,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNEVAL_H_ */
