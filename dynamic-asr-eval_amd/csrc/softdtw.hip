// Soft-DTW on the MI355X — replaces the reference's only GPU kernels, the two Numba @cuda.jit wavefront kernels
// `compute_softdtw_cuda` / `compute_softdtw_backward_cuda` (reference wav2vec2/soft_dtw_cuda.py:33-111), their
// autograd wrapper (:114-175) and the broadcast distance `_euclidean_dist_func` (:319-329; materialises [B,N,M,d]).
//
//   forward   R[i,j] = D[i-1,j-1] + softmin_gamma(R[i-1,j-1], R[i-1,j], R[i,j-1])      (max-shifted log-sum-exp)
//   backward  E[i,j] = E[i+1,j] a + E[i,j+1] b + E[i+1,j+1] c,  a = exp((R[i+1,j] - R[i,j] - D[i+1,j]) / gamma) ...
// with the reference's padding and boundary values (R is [B, N+2, M+2] with a +inf border, R[0,0] = 0; the backward
// sets the last row/column to -inf, R[N+1,M+1] = R[N,M], E[N+1,M+1] = 1, +-inf cells are reset to -inf) and its
// Sakoe-Chiba rule (cells with |i - j| > bandwidth > 0 are skipped).
//
// Numerics: the lattice (R, and E inside the scan) is kept in fp64 — the reference's CPU path is fp64 too
// (soft_dtw_cuda.py:189,214) while its CUDA path inherits fp32 from D; at |R| ~ 2e3 an fp32 lattice loses ~1 % of the
// gradient.  The scan is latency-bound, so fp64 costs bytes (R workspace is 8 B/cell), not time.  D, value, E are fp32.
//
// MI355X mapping: one workgroup per sequence pair walks the anti-diagonals; the three live diagonals of R (forward)
// / E (backward) rotate through LDS so each cell costs 3 LDS reads instead of 3 dependent global loads, one barrier
// per diagonal.  A thread owns rows i = tid, tid + blockDim, ... so N and M are NOT limited to 1024 (the reference falls
// back to the CPU beyond that, soft_dtw_cuda.py:312-314).  The pairwise distance never materialises [B,N,M,d].
#include "common.h"

namespace {

constexpr int TPB = 256;

// D[b,i,j] = sum_k (x[b,i,k] - y[b,j,k])^2
__global__ __launch_bounds__(TPB) void sqdist_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                     float* __restrict__ D, int N, int M, int d) {
    const int64_t b = blockIdx.z;
    const int i = blockIdx.y;
    const float* xi = x + (b * N + i) * d;
    for (int j = blockIdx.x * TPB + threadIdx.x; j < M; j += gridDim.x * TPB) {
        const float* yj = y + (b * M + j) * d;
        float s = 0.f;
        for (int k = 0; k < d; ++k) {
            const float t = xi[k] - yj[k];
            s += t * t;
        }
        D[(b * N + i) * M + j] = s;
    }
}

// dx[b,i,k] = sum_j 2 * G[b,i,j] * (x[b,i,k] - y[b,j,k])      (G = upstream gradient w.r.t. D)
__global__ __launch_bounds__(TPB) void sqdist_bwd_x_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ G, float* __restrict__ dx, int N, int M,
                                                           int d) {
    const int64_t b = blockIdx.y;
    const int i = blockIdx.x;
    const float* g = G + (b * N + i) * M;
    for (int k = threadIdx.x; k < d; k += TPB) {
        const float xv = x[(b * N + i) * d + k];
        float s = 0.f;
        for (int j = 0; j < M; ++j) s += g[j] * (xv - y[(b * M + j) * d + k]);
        dx[(b * N + i) * d + k] = 2.f * s;
    }
}

__device__ __forceinline__ double softmin3(double r0, double r1, double r2, double gamma) {
    // reference :66-71 — r* are -R/gamma
    const double rmax = fmax(fmax(r0, r1), r2);
    const double rsum = exp(r0 - rmax) + exp(r1 - rmax) + exp(r2 - rmax);
    return -gamma * (log(rsum) + rmax);
}

// Forward wavefront.  R [B, N+2, M+2] must be pre-filled by the kernel itself (border +inf, R[0,0] = 0).
__global__ __launch_bounds__(1024) void softdtw_fwd_kernel(const float* __restrict__ D, double* __restrict__ R,
                                                            float* __restrict__ value, int N, int M, float gamma_f,
                                                            float bandwidth) {
    extern __shared__ __attribute__((aligned(16))) double diag[];  // 3 x (N + 1): R on diagonals p-2, p-1, p by row index i
    const double gamma = (double)gamma_f;
    const int64_t b = blockIdx.x;
    const float* Db = D + b * (int64_t)N * M;
    double* Rb = R + b * (int64_t)(N + 2) * (M + 2);
    const int W = M + 2, LN = N + 1;
    const double inv_gamma = 1.0 / gamma;
    // initialise R: everything +inf, R[0,0] = 0
    for (int64_t e = threadIdx.x; e < (int64_t)(N + 2) * W; e += blockDim.x) Rb[e] = INFINITY;
    double* d2 = diag;           // diagonal p-2  (q = i + j, 1-based cells: q = p + 2)
    double* d1 = diag + LN;      // diagonal p-1
    double* d0 = diag + 2 * LN;  // diagonal p (being written)
    // 1-based (i, j); diagonal q = i + j.  q = 0: only R[0,0] = 0; q = 1: R[0,1] = R[1,0] = inf.
    for (int i = threadIdx.x; i <= N; i += blockDim.x) {
        d2[i] = (i == 0) ? 0.0 : (double)INFINITY;  // q = 0: cell (i, -i) exists only for i = 0
        d1[i] = INFINITY;                            // q = 1
    }
    __syncthreads();
    if (threadIdx.x == 0) Rb[0] = 0.0;
    for (int q = 2; q <= N + M; ++q) {
        for (int i = threadIdx.x; i <= N; i += blockDim.x) {
            const int j = q - i;
            double v = INFINITY;
            if (i >= 1 && j >= 1 && j <= M) {
                if (!(bandwidth > 0.f && fabsf((float)(i - j)) > bandwidth)) {
                    const double r0 = -d2[i - 1] * inv_gamma;  // R[i-1, j-1]
                    const double r1 = -d1[i - 1] * inv_gamma;  // R[i-1, j]
                    const double r2 = -d1[i] * inv_gamma;      // R[i, j-1]
                    v = (double)Db[(int64_t)(i - 1) * M + (j - 1)] + softmin3(r0, r1, r2, gamma);
                    Rb[(int64_t)i * W + j] = v;
                }
            } else if (i == 0 && j == 0) {
                v = 0.0;
            }
            d0[i] = v;
        }
        __syncthreads();
        double* t = d2; d2 = d1; d1 = d0; d0 = t;
    }
    if (threadIdx.x == 0) value[b] = (float)Rb[(int64_t)N * W + M];
}

// Backward wavefront.  Reads R as left by the forward (finite inside, +inf border), applies the reference's boundary
// rewrite on the fly, writes E [B, N, M] (gradient of the value w.r.t. D).
__global__ __launch_bounds__(1024) void softdtw_bwd_kernel(const float* __restrict__ D, const double* __restrict__ R,
                                                            float* __restrict__ E, int N, int M, float gamma, float bandwidth) {
    extern __shared__ __attribute__((aligned(16))) double diag[];  // 3 x (N + 2): E on diagonals q+2, q+1, q by row index i
    const int64_t b = blockIdx.x;
    const float* Db = D + b * (int64_t)N * M;
    const double* Rb = R + b * (int64_t)(N + 2) * (M + 2);
    float* Eb = E + b * (int64_t)N * M;
    const int W = M + 2, LN = N + 2;
    const double inv_gamma = 1.0 / (double)gamma;
    const double RNM = Rb[(int64_t)N * W + M];
    // R' = R with: last row / column -inf, R'[N+1, M+1] = R[N, M], and +-inf interior cells -> -inf (reference :161-163,:100-101)
    auto Rv = [&](int i, int j) -> double {
        if (i == N + 1 && j == M + 1) return RNM;
        if (i == N + 1 || j == M + 1) return -(double)INFINITY;
        const double r = Rb[(int64_t)i * W + j];
        return isinf(r) ? -(double)INFINITY : r;
    };
    auto Dv = [&](int i, int j) -> double {  // padded D_: zero outside [1..N] x [1..M]
        return (i >= 1 && i <= N && j >= 1 && j <= M) ? (double)Db[(int64_t)(i - 1) * M + (j - 1)] : 0.0;
    };
    double* e2 = diag;           // diagonal q+2
    double* e1 = diag + LN;      // diagonal q+1
    double* e0 = diag + 2 * LN;  // diagonal q
    // q = N + M + 2 holds only E[N+1, M+1] = 1; q = N + M + 1 holds E[N+1, M] = E[N, M+1] = 0.
    for (int i = threadIdx.x; i <= N + 1; i += blockDim.x) {
        e2[i] = (i == N + 1) ? 1.0 : 0.0;
        e1[i] = 0.0;
    }
    __syncthreads();
    for (int q = N + M; q >= 2; --q) {
        for (int i = threadIdx.x; i <= N + 1; i += blockDim.x) {
            const int j = q - i;
            double v = 0.0;
            if (i >= 1 && i <= N && j >= 1 && j <= M) {
                if (!(bandwidth > 0.f && fabsf((float)(i - j)) > bandwidth)) {
                    const double rij = Rv(i, j);
                    const double a = exp((Rv(i + 1, j) - rij - Dv(i + 1, j)) * inv_gamma);
                    const double bb = exp((Rv(i, j + 1) - rij - Dv(i, j + 1)) * inv_gamma);
                    const double c = exp((Rv(i + 1, j + 1) - rij - Dv(i + 1, j + 1)) * inv_gamma);
                    v = e1[i + 1] * a + e1[i] * bb + e2[i + 1] * c;  // E[i+1,j], E[i,j+1], E[i+1,j+1]
                }
                Eb[(int64_t)(i - 1) * M + (j - 1)] = (float)v;
            }
            e0[i] = v;
        }
        __syncthreads();
        double* t = e2; e2 = e1; e1 = e0; e0 = t;
    }
}

inline int scan_threads(int n) {
    int t = (n + 63) / 64 * 64;
    if (t > 1024) t = 1024;
    if (t < 64) t = 64;
    return t;
}

}  // namespace

extern "C" int dyn_sqdist(const float* x, const float* y, float* D, int64_t B, int64_t N, int64_t M, int64_t d, void* stream) {
    DYN_REQUIRE(x && y && D && B >= 0 && N > 0 && M > 0 && d > 0 && B < 65536 && N < 65536, DYN_E_ARG, "dyn_sqdist: bad arguments");
    if (B == 0) return DYN_OK;
    hipLaunchKernelGGL(sqdist_kernel, dim3((unsigned)dyn::cdiv(M, TPB), (unsigned)N, (unsigned)B), dim3(TPB), 0, (hipStream_t)stream,
                       x, y, D, (int)N, (int)M, (int)d);
    return dyn::check_launch("dyn_sqdist");
}

extern "C" int dyn_sqdist_bwd_x(const float* x, const float* y, const float* G, float* dx, int64_t B, int64_t N, int64_t M,
                                int64_t d, void* stream) {
    DYN_REQUIRE(x && y && G && dx && B >= 0 && N > 0 && M > 0 && d > 0 && B < 65536, DYN_E_ARG, "dyn_sqdist_bwd_x: bad arguments");
    if (B == 0) return DYN_OK;
    hipLaunchKernelGGL(sqdist_bwd_x_kernel, dim3((unsigned)N, (unsigned)B), dim3(TPB), 0, (hipStream_t)stream, x, y, G, dx, (int)N,
                       (int)M, (int)d);
    return dyn::check_launch("dyn_sqdist_bwd_x");
}

extern "C" int dyn_softdtw_fwd(const float* D, double* R, float* value, int64_t B, int64_t N, int64_t M, float gamma,
                               float bandwidth, void* stream) {
    DYN_REQUIRE(D && R && value && B >= 0 && N > 0 && M > 0 && gamma > 0.f, DYN_E_ARG, "dyn_softdtw_fwd: bad arguments");
    DYN_REQUIRE(3 * (N + 2) * 8 <= 160 * 1024 - 1024, DYN_E_UNSUPPORTED, "dyn_softdtw_fwd: N=%lld exceeds the LDS diagonal buffers",
                (long long)N);
    if (B == 0) return DYN_OK;
    const size_t shm = (size_t)3 * (N + 1) * sizeof(double);
    if (shm > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)softdtw_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(softdtw_fwd_kernel, dim3((unsigned)B), dim3(scan_threads((int)N + 1)), shm, (hipStream_t)stream, D, R, value,
                       (int)N, (int)M, gamma, bandwidth);
    return dyn::check_launch("dyn_softdtw_fwd");
}

extern "C" int dyn_softdtw_bwd(const float* D, const double* R, float* E, int64_t B, int64_t N, int64_t M, float gamma,
                               float bandwidth, void* stream) {
    DYN_REQUIRE(D && R && E && B >= 0 && N > 0 && M > 0 && gamma > 0.f, DYN_E_ARG, "dyn_softdtw_bwd: bad arguments");
    DYN_REQUIRE(3 * (N + 2) * 8 <= 160 * 1024 - 1024, DYN_E_UNSUPPORTED, "dyn_softdtw_bwd: N=%lld exceeds the LDS diagonal buffers",
                (long long)N);
    if (B == 0) return DYN_OK;
    const size_t shm = (size_t)3 * (N + 2) * sizeof(double);
    if (shm > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)softdtw_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(softdtw_bwd_kernel, dim3((unsigned)B), dim3(scan_threads((int)N + 2)), shm, (hipStream_t)stream, D, R, E, (int)N,
                       (int)M, gamma, bandwidth);
    return dyn::check_launch("dyn_softdtw_bwd");
}
