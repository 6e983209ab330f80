// Face alignment warp: frame + one 2x3 similarity matrix per face -> 112x112 uint8 crops, the
// step between the detector and the embedding network (Mobilefacenet/utils/align_trans.py:50-52 of
// the reference: cv2.warpAffine(img, M[:2], (W, H))[0:112, 0:112]).
//
// Only the 112x112 window the reference keeps is computed (the reference warps the whole frame and
// throws the rest away), for all faces of a frame in one launch, straight into the uint8 HWC batch
// the embedding network's first kernel reads.
//
// Arithmetic is OpenCV's for 8-bit INTER_LINEAR with a constant (0) border:
//   * M is inverted in double precision;
//   * source coordinates are integers with 10 fractional bits, X = rint(a x 1024) + rint((b y + c) 1024) + 16,
//     cut to 1/32 pixel (X >> 5): integer part X >> 10, fraction fx = (X >> 5) & 31;
//   * the four taps are weighted by (32-fx)(32-fy) 32, fx (32-fy) 32, (32-fx) fy 32, fx fy 32 (16-bit, the
//     first saturating at 32767) and the sum is rounded as (sum + 2^14) >> 15; taps outside the frame read 0.
// Every double operation is a separate IEEE operation (no contraction), so the integer coordinates
// equal those of a host computation with the same formulas.
#include "common.h"

#pragma clang fp contract(off)

namespace fra {

constexpr int FA_PX_PER_THREAD = 4;

struct AffineInv { double a, b, c, d, e, f; };      // src_x = a x + b y + c,  src_y = d x + e y + f

__device__ __forceinline__ AffineInv invert_affine(const double* __restrict__ M) {
    double m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3], m4 = M[4], m5 = M[5];
    double D = m0 * m4 - m1 * m3;
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = m4 * D, A22 = m0 * D;
    m0 = A11; m1 = m1 * -D; m3 = m3 * -D; m4 = A22;
    const double b1 = -m0 * m2 - m1 * m5;
    const double b2 = -m3 * m2 - m4 * m5;
    return {m0, m1, b1, m3, m4, b2};
}

__device__ __forceinline__ long long round_ll(double v) {
    v = fmin(fmax(v, -9.0e18), 9.0e18);
    return (long long)rint(v);                       // round half to even
}

// frame u8 [H,W,3]; M f64 [F][6]; crops u8 [F,OH,OW,3].  One thread = 4 neighbouring output pixels (12 bytes).
__global__ __launch_bounds__(256) void align_warp_kernel(const uint8_t* __restrict__ frame, int H, int W,
                                                         const double* __restrict__ M, uint8_t* __restrict__ crops,
                                                         int OH, int OW) {
    const int face = blockIdx.y;
    const int groups = OW / FA_PX_PER_THREAD;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= OH * groups) return;
    const int y = t / groups, x0 = (t - y * groups) * FA_PX_PER_THREAD;
    const AffineInv m = invert_affine(M + (size_t)face * 6);
    const long long X0 = round_ll((m.b * (double)y + m.c) * 1024.0) + 16;
    const long long Y0 = round_ll((m.e * (double)y + m.f) * 1024.0) + 16;
    uint32_t packed[3] = {0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < FA_PX_PER_THREAD; ++i) {
        const double x = (double)(x0 + i);
        const long long X = (X0 + round_ll(m.a * x * 1024.0)) >> 5;
        const long long Y = (Y0 + round_ll(m.d * x * 1024.0)) >> 5;
        long long sxl = X >> 5, syl = Y >> 5;
        sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);
        syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
        const int sx = (int)sxl, sy = (int)syl;
        const int fx = (int)(X & 31), fy = (int)(Y & 31);
        int w00 = (32 - fx) * (32 - fy) * 32;
        w00 = w00 > 32767 ? 32767 : w00;
        const int w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        const bool x_lo = sx >= 0 && sx < W, x_hi = sx + 1 >= 0 && sx + 1 < W;
        const bool y_lo = sy >= 0 && sy < H, y_hi = sy + 1 >= 0 && sy + 1 < H;
        const int cx0 = min(max(sx, 0), W - 1), cx1 = min(max(sx + 1, 0), W - 1);
        const int cy0 = min(max(sy, 0), H - 1), cy1 = min(max(sy + 1, 0), H - 1);
        const uint8_t* p00 = frame + ((size_t)cy0 * W + cx0) * 3;
        const uint8_t* p01 = frame + ((size_t)cy0 * W + cx1) * 3;
        const uint8_t* p10 = frame + ((size_t)cy1 * W + cx0) * 3;
        const uint8_t* p11 = frame + ((size_t)cy1 * W + cx1) * 3;
        const int k00 = (y_lo && x_lo) ? w00 : 0, k01 = (y_lo && x_hi) ? w01 : 0;
        const int k10 = (y_hi && x_lo) ? w10 : 0, k11 = (y_hi && x_hi) ? w11 : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int acc = (int)p00[c] * k00 + (int)p01[c] * k01 + (int)p10[c] * k10 + (int)p11[c] * k11;
            const uint32_t v = (uint32_t)min(max((acc + (1 << 14)) >> 15, 0), 255);
            const int byte = i * 3 + c;
            packed[byte >> 2] |= v << (8 * (byte & 3));
        }
    }
    uint32_t* out = reinterpret_cast<uint32_t*>(crops + (((size_t)face * OH + y) * OW + x0) * 3);   // 12-byte aligned
    out[0] = packed[0];
    out[1] = packed[1];
    out[2] = packed[2];
}

}  // namespace fra

using namespace fra;

extern "C" int fa_align_warp(const uint8_t* frame, int H, int W, const double* matrices, int n_faces, uint8_t* crops,
                             int out_h, int out_w, fra_stream stream) {
    FRA_REQUIRE(H > 0 && W > 0 && H <= 32767 && W <= 32767, FRA_ERR_ARG, "fa_align_warp: frame %d x %d (1..32767 each)", H, W);
    FRA_REQUIRE(n_faces >= 0 && n_faces <= 65535, FRA_ERR_ARG, "fa_align_warp: %d faces (0..65535)", n_faces);
    FRA_REQUIRE(out_h > 0 && out_w > 0 && out_w % FA_PX_PER_THREAD == 0 && out_h <= 4096 && out_w <= 4096, FRA_ERR_ARG,
                "fa_align_warp: crop %d x %d (width a multiple of %d, both <= 4096)", out_h, out_w, FA_PX_PER_THREAD);
    if (n_faces == 0) return FRA_OK;                 // nothing detected in this frame: no pointer is touched
    FRA_REQUIRE(frame && crops && matrices, FRA_ERR_ARG, "fa_align_warp: null pointer");
    FRA_REQUIRE(((uintptr_t)crops & 3) == 0 && ((uintptr_t)matrices & 7) == 0, FRA_ERR_ARG,
                "fa_align_warp: crops must be 4-byte aligned, matrices 8-byte aligned");
    const int threads = out_h * (out_w / FA_PX_PER_THREAD);
    hipLaunchKernelGGL(align_warp_kernel, dim3((threads + 255) / 256, n_faces), dim3(256), 0, (hipStream_t)stream, frame, H,
                       W, matrices, crops, out_h, out_w);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}
