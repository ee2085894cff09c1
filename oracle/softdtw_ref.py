"""ORACLE (test infrastructure only — never imported by the product path).

numpy float64 restatement of the reference's soft-DTW recurrences, vectorised over the batch:
  forward   R[i,j] = D[i-1,j-1] + softmin_gamma(R[i-1,j-1], R[i-1,j], R[i,j-1])   reference wav2vec2/soft_dtw_cuda.py:184-206
  backward  E[i,j] = E[i+1,j] a + E[i,j+1] b + E[i+1,j+1] c                        reference wav2vec2/soft_dtw_cuda.py:209-239
with the same padding / boundary initialisation (R is [B, N+2, M+2], +inf border, R[0,0] = 0; backward sets the last
row/column to -inf, R[-1,-1] = R[-2,-2], E[-1,-1] = 1) and the same Sakoe-Chiba pruning rule (`0 < bandwidth < |i-j|`).
`sqdist` restates `SoftDTW._euclidean_dist_func` (soft_dtw_cuda.py:319-329).
The reference's own file cannot be imported here (numba is absent); the only pin the reference holds for this code is
its CPU<->GPU allclose self-check (soft_dtw_cuda.py:382-428), which tests/test_softdtw_gpu.py re-runs between this
restatement and the HIP kernels at the same shapes and tolerances."""
import numpy as np


def sqdist(x, y):
    """x [B, N, d], y [B, M, d] -> D [B, N, M] = sum_d (x_i - y_j)^2"""
    x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    return ((x[:, :, None, :] - y[:, None, :, :]) ** 2).sum(-1)


def softdtw_forward(D, gamma, bandwidth=0.0):
    D = np.asarray(D, dtype=np.float64)
    B, N, M = D.shape
    R = np.full((B, N + 2, M + 2), np.inf)
    R[:, 0, 0] = 0.0
    for j in range(1, M + 1):
        for i in range(1, N + 1):
            if 0 < bandwidth < abs(i - j):
                continue
            r0 = -R[:, i - 1, j - 1] / gamma
            r1 = -R[:, i - 1, j] / gamma
            r2 = -R[:, i, j - 1] / gamma
            rmax = np.maximum(np.maximum(r0, r1), r2)
            rsum = np.exp(r0 - rmax) + np.exp(r1 - rmax) + np.exp(r2 - rmax)
            R[:, i, j] = D[:, i - 1, j - 1] + (-gamma * (np.log(rsum) + rmax))
    return R


def softdtw_backward(D_, R, gamma, bandwidth=0.0):
    D_ = np.asarray(D_, dtype=np.float64)
    B, N, M = D_.shape
    R = R.copy()
    D = np.zeros((B, N + 2, M + 2)); E = np.zeros((B, N + 2, M + 2))
    D[:, 1:N + 1, 1:M + 1] = D_
    E[:, -1, -1] = 1.0
    R[:, :, -1] = -np.inf
    R[:, -1, :] = -np.inf
    R[:, -1, -1] = R[:, -2, -2]
    with np.errstate(invalid="ignore", over="ignore"):
        for j in range(M, 0, -1):
            for i in range(N, 0, -1):
                inf = np.isinf(R[:, i, j])
                R[inf, i, j] = -np.inf
                if 0 < bandwidth < abs(i - j):
                    continue
                a = np.exp((R[:, i + 1, j] - R[:, i, j] - D[:, i + 1, j]) / gamma)
                b = np.exp((R[:, i, j + 1] - R[:, i, j] - D[:, i, j + 1]) / gamma)
                c = np.exp((R[:, i + 1, j + 1] - R[:, i, j] - D[:, i + 1, j + 1]) / gamma)
                E[:, i, j] = E[:, i + 1, j] * a + E[:, i, j + 1] * b + E[:, i + 1, j + 1] * c
    return E[:, 1:N + 1, 1:M + 1]


def softdtw_forward_backward(D, gamma, bandwidth=0.0):
    """-> (value [B] = R[:, N, M], grad wrt D [B, N, M] for grad_output = 1)"""
    R = softdtw_forward(D, gamma, bandwidth)
    return R[:, -2, -2].copy(), softdtw_backward(D, R, gamma, bandwidth)
