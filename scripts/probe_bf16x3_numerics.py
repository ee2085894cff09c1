"""CPU-only numerics study for a next-round lever: an fp32 GEMM computed on the bf16 matrix cores by splitting every fp32
operand into three bf16 terms (a = a0 + a1 + a2 exactly: 3 x 8 mantissa bits) and keeping the six products a_i b_j with i + j <= 2, each
accumulated in fp32 as the MFMA does.  Dense bf16 MFMA is ~2.5 PFLOP/s on MI355X against 157 TFLOP/s for v_mfma_f32_32x32x2_f32, so six bf16
products per fp32 product have a ceiling of ~417 TFLOP/s fp32-equivalent.  The question this script answers: how far is such a product from the
float64 result, next to a plain fp32 FMA-chain GEMM (what dyn_gemm_f32 computes, up to summation order)?

Emulation: bf16 terms by round-to-nearest-even on the fp32 bit pattern; a product of two bf16 values is exact in fp32 (16 significant bits), so the
only roundings are the fp32 accumulations, done here k by k in float32 like a matrix core's accumulator.
    python scripts/probe_bf16x3_numerics.py [out.json]"""
import json
import sys

import numpy as np


def bf16_round(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    a0 = bf16_round(x)
    r1 = (x - a0).astype(np.float32)         # exact: a0 is x's leading bits
    a1 = bf16_round(r1)
    r2 = (r1 - a1).astype(np.float32)
    a2 = bf16_round(r2)
    return a0, a1, a2, float(np.abs(r2 - a2).max())


def gemm_fp32_chain(A, B):
    """sum_k a_ik b_kj with one fp32 rounding per step (FMA: the product is not rounded)."""
    M, K = A.shape
    acc = np.zeros((M, B.shape[1]), np.float32)
    for k in range(K):
        acc = (acc.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * B[k:k + 1, :].astype(np.float64)).astype(np.float32)
    return acc


def gemm_bf16x3(A, B, terms):
    As, Bs = split3(A)[:3], split3(B)[:3]
    M, K = A.shape
    acc = np.zeros((M, B.shape[1]), np.float32)
    # small terms first within a k step, as a kernel would order the six MFMAs of a K block
    for k in range(K):
        for i, j in terms:
            p = As[i][:, k:k + 1].astype(np.float64) * Bs[j][k:k + 1, :].astype(np.float64)     # exact in fp32
            acc = (acc.astype(np.float64) + p).astype(np.float32)
    return acc


def main():
    rng = np.random.default_rng(0)
    out = {"what": __doc__.split("\n\n")[0], "cases": []}
    six = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]
    three = [(1, 0), (0, 1), (0, 0)]
    nine = [(2, 2), (2, 1), (1, 2)] + six
    for name, M, N, K, scale in (("activations x weights, K = 768", 64, 64, 768, 0.05), ("K = 3072", 48, 48, 3072, 0.02),
                                 ("attention scores, K = 128", 64, 64, 128, 1.0), ("weight gradient, K = 4096 frames", 32, 32, 4096, 0.01)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = (rng.standard_normal((K, N)) * scale).astype(np.float32)
        ref = A.astype(np.float64) @ B.astype(np.float64)
        nrm = float(np.abs(ref).max())
        row = {"case": name, "M": M, "N": N, "K": K, "split_residual_max": split3(A)[3]}
        for tag, got in (("fp32_fma_chain", gemm_fp32_chain(A, B)), ("bf16x3_six_products", gemm_bf16x3(A, B, six)),
                         ("bf16x3_nine_products", gemm_bf16x3(A, B, nine)), ("bf16x2_three_products", gemm_bf16x3(A, B, three))):
            err = np.abs(got.astype(np.float64) - ref)
            row[tag] = {"max_abs_err_over_max_abs_ref": float(err.max() / nrm), "rms_err_over_max_abs_ref": float(np.sqrt((err ** 2).mean()) / nrm)}
        out["cases"].append(row)
        print(json.dumps(row), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
