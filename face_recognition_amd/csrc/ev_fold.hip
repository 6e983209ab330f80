// Verification protocol on the device: the per-fold centring, normalisation and pair scoring of
// evaluation_10_fold (Evaluation.py:83-94 of the reference) with the features resident in HBM.
//
// The reference does this in numpy, and its results (a threshold picked on a 1e-4 grid, accuracies
// that are counts) are discrete functions of the scores, so the scores are reproduced BIT FOR BIT:
// every float32 operation is issued in numpy's order with explicit round-to-nearest intrinsics
// (no fused multiply-add, no reassociation).
//   * np.mean(.., axis 0) of a C-contiguous array adds the rows one after another per column and
//     divides by the row count                                              -> col_mean_kernel
//   * np.sum(.., axis 1) over 512 contiguous float32 is numpy's pairwise sum: blocks of 128, inside
//     a block eight interleaved accumulators r[k] += a[8 m + k] (m ascending), combined as
//     ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), blocks combined as a balanced binary tree
//                                                                           -> eight lanes per row
// The threshold search and the accuracy count then run on the host over the P scores of a fold.
#include "common.h"

// No fused multiply-add anywhere in this file: numpy rounds the product before the sum.  The
// arithmetic below is written with plain operators for that reason -- the __fmul_rn / __fadd_rn
// helpers are inline functions compiled under the header's (contracting) mode and fuse after inlining.
#pragma clang fp contract(off)

namespace fra {

// mu[j] = (sum over selected rows of fl, then of fr, in row order) / (2 * n_sel); one thread per column
__global__ __launch_bounds__(64) void col_mean_kernel(const float* __restrict__ fl, const float* __restrict__ fr,
                                                      const unsigned char* __restrict__ sel, int P, int D, int n_sel,
                                                      float* __restrict__ mu) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= D) return;
    float acc = 0.0f;
    for (int pass = 0; pass < 2; ++pass) {
        const float* a = pass == 0 ? fl : fr;
        // an unselected row contributes +0.0, which leaves the running sum bit for bit as it was, so the
        // loop is branch free and its loads can be issued well ahead of the (sequential) additions
#pragma unroll 16
        for (int i = 0; i < P; ++i) {
            const float v = a[(size_t)i * D + j];
            acc = acc + (sel[i] ? v : 0.0f);
        }
    }
    mu[j] = acc / (float)(2 * n_sel);
}

// correctly rounded float32 square root (numpy's np.sqrt): the float32 hardware instruction is good to
// 1 ulp only; the double-precision root rounded once more to float32 is exact for float32 inputs
__device__ __forceinline__ float sqrt_f32_exact(float x) { return (float)sqrt((double)x); }

// numpy's pairwise sum of one row spread over 8 lanes: lane k holds v[b][m] = element 128 b + 8 m + k
template <int NBLK>
__device__ __forceinline__ float pairwise_row_sum(const float (&v)[NBLK][16]) {
    float blk[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        float r = v[b][0];
#pragma unroll
        for (int m = 1; m < 16; ++m) r = r + v[b][m];
        // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)): an xor butterfly over the 8 lanes (float add commutes bitwise)
        r = r + __shfl_xor(r, 1, 64);
        r = r + __shfl_xor(r, 2, 64);
        r = r + __shfl_xor(r, 4, 64);
        blk[b] = r;
    }
#pragma unroll
    for (int w = 1; w < NBLK; w *= 2)
#pragma unroll
        for (int b = 0; b < NBLK; b += 2 * w) blk[b] = blk[b] + blk[b + w];
    return blk[0];
}

// per row: x = f - mu; x /= sqrt(sum x^2) (both sides, written back); score = sum (xl - xr)^2 or sum xl * xr
template <int NBLK>
__global__ __launch_bounds__(256) void centre_norm_score_kernel(float* __restrict__ fl, float* __restrict__ fr,
                                                                const float* __restrict__ mu, int P, int method,
                                                                float* __restrict__ scores) {
    constexpr int D = NBLK * 128;
    const int k = threadIdx.x & 7;
    const int row = blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool live = row < P;                  // whole 8-lane groups are live or not; shuffles stay inside a group
    const size_t base = (size_t)(live ? row : 0) * D;
    float xl[NBLK][16], xr[NBLK][16], t[NBLK][16];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int e = 128 * b + 8 * m + k;
            const float c = mu[e];
            xl[b][m] = fl[base + e] - c;
            xr[b][m] = fr[base + e] - c;
        }
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int m = 0; m < 16; ++m) t[b][m] = xl[b][m] * xl[b][m];
    const float nl = sqrt_f32_exact(pairwise_row_sum<NBLK>(t));
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int m = 0; m < 16; ++m) t[b][m] = xr[b][m] * xr[b][m];
    const float nr = sqrt_f32_exact(pairwise_row_sum<NBLK>(t));
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            xl[b][m] = xl[b][m] / nl;
            xr[b][m] = xr[b][m] / nr;
            if (method == 0) {
                const float d = xl[b][m] - xr[b][m];
                t[b][m] = d * d;
            } else {
                t[b][m] = xl[b][m] * xr[b][m];
            }
        }
    const float s = pairwise_row_sum<NBLK>(t);
    if (live) {
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int e = 128 * b + 8 * m + k;
                fl[base + e] = xl[b][m];
                fr[base + e] = xr[b][m];
            }
        if (k == 0) scores[row] = s;
    }
}

}  // namespace fra

using namespace fra;

extern "C" int ev_fold_scores(float* fl, float* fr, const uint8_t* val_mask, int n_val, int P, int D, int method,
                              float* mu_scratch, float* scores_out, fra_stream stream) {
    FRA_REQUIRE(fl && fr && val_mask && mu_scratch && scores_out, FRA_ERR_ARG, "ev_fold_scores: null pointer");
    FRA_REQUIRE(P > 0 && n_val > 0 && n_val <= P, FRA_ERR_ARG, "ev_fold_scores: P=%d n_val=%d", P, n_val);
    FRA_REQUIRE(D == 128 || D == 256 || D == 512, FRA_ERR_ARG, "ev_fold_scores: feature width %d not in {128,256,512}", D);
    FRA_REQUIRE(method == 0 || method == 1, FRA_ERR_ARG, "ev_fold_scores: method %d (0 = l2_distance, 1 = cos_distance)",
                method);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(col_mean_kernel, dim3((D + 63) / 64), dim3(64), 0, s, fl, fr, val_mask, P, D, n_val, mu_scratch);
    FRA_HIP(hipGetLastError());
    const dim3 grid((P + 31) / 32), block(256);
    switch (D) {
        case 128: hipLaunchKernelGGL(centre_norm_score_kernel<1>, grid, block, 0, s, fl, fr, mu_scratch, P, method, scores_out); break;
        case 256: hipLaunchKernelGGL(centre_norm_score_kernel<2>, grid, block, 0, s, fl, fr, mu_scratch, P, method, scores_out); break;
        default: hipLaunchKernelGGL(centre_norm_score_kernel<4>, grid, block, 0, s, fl, fr, mu_scratch, P, method, scores_out); break;
    }
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}
