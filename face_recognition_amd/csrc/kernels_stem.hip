// Fused network head for gfx950: Conv2d(3,64,3x3,s2)+BN+PReLU (face_model.py:80) followed by the
// depthwise 3x3 Conv_block (face_model.py:81) in one kernel.
//
// Same structure as the fused block (kernels_block.hip): a workgroup owns TR output rows of one
// image; the stem conv runs on the matrix cores as an im2col GEMM (K = 27 taps padded to 32, the
// patch fragments gathered straight from the caller's fp32 NCHW image), its PReLU output goes into
// an LDS image [row][x+1][64 ch] with zero borders, and every lane then computes the depthwise 3x3
// for its pixel from nine 16-byte LDS reads per 8 channels and stores NHWC.  The 56x56x64 stem
// output never touches HBM.
//
// The stem weight matrix is stored with its rows (output channels) permuted by perm_pos(): MFMA
// row 16g+8a+4h+i computes output channel 16g+8h+4a+i, so the 8 values a lane holds per
// 16-channel group are 8 CONSECUTIVE channels and travel as one 16-byte unit.
#include "common.h"

namespace fra {

__device__ __forceinline__ int eswz2(int slot) { return (slot >> 1) & 7; }   // as eswz() in kernels_block.hip

template <typename T, int TR, int NW, int MINW>
__global__ __launch_bounds__(NW * 64, MINW)
void stemdw_kernel(const float* __restrict__ X, const T* __restrict__ WA, const float* __restrict__ b1,
                   const float* __restrict__ s1, const T* __restrict__ W2, const float* __restrict__ b2,
                   const float* __restrict__ s2, T* __restrict__ O) {
    typedef typename vec8<T>::type V8;
    constexpr int NT = NW * 64;
    constexpr int H = 56, W = 56, HI = 112;
    constexpr int RIN = TR + 2, PITCH = W + 2;
    constexpr int PIN = RIN * W, POUT = TR * W;
    constexpr int MTI = (PIN + 31) / 32, MTO = (POUT + 31) / 32;
    constexpr int MTIW = (MTI + NW - 1) / NW, MTOW = (MTO + NW - 1) / NW;
    constexpr int ES_BYTES = RIN * PITCH * 128;
    constexpr int OFF_W2 = ES_BYTES, OFF_B1 = OFF_W2 + 9 * 64 * 2, OFF_S1 = OFF_B1 + 256, OFF_B2 = OFF_S1 + 256,
                  OFF_S2 = OFF_B2 + 256;
    static_assert(H % TR == 0, "band height");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const es = lds;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int band = blockIdx.x, img = blockIdx.y;
    const int y0 = band * TR, row0 = y0 - 1;

    for (int i = tid; i < ES_BYTES / 16; i += NT) reinterpret_cast<f32x4*>(es)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 9 * 64 / 8; i += NT) reinterpret_cast<V8*>(lds + OFF_W2)[i] = reinterpret_cast<const V8*>(W2)[i];
    if (tid < 16) {
        reinterpret_cast<f32x4*>(lds + OFF_B1)[tid] = reinterpret_cast<const f32x4*>(b1)[tid];
        reinterpret_cast<f32x4*>(lds + OFF_S1)[tid] = reinterpret_cast<const f32x4*>(s1)[tid];
        reinterpret_cast<f32x4*>(lds + OFF_B2)[tid] = reinterpret_cast<const f32x4*>(b2)[tid];
        reinterpret_cast<f32x4*>(lds + OFF_S2)[tid] = reinterpret_cast<const f32x4*>(s2)[tid];
    }

    // stem weights as MFMA A fragments: 2 tiles of 32 (permuted) output channels x 2 k-steps
    V8 wa[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int s = 0; s < 2; ++s) wa[m][s] = *reinterpret_cast<const V8*>(WA + (32 * m + r) * 32 + 16 * s + 8 * h);

    // the 16 taps this lane gathers: t = 16 s + 8 h + j -> (c, ky, kx); offset inside the 3x112x112 image
    int toff[2][8];
    unsigned ky0 = 0, kx0 = 0, live = 0;   // bit 8 s + j
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = 16 * s + 8 * h + j;
            const int c = t / 9, rem = t - 9 * c, ky = rem / 3, kx = rem - 3 * ky;
            toff[s][j] = c * HI * HI + ky * HI + kx;
            if (ky == 0) ky0 |= 1u << (8 * s + j);
            if (kx == 0) kx0 |= 1u << (8 * s + j);
            if (t < 27) live |= 1u << (8 * s + j);
        }
    __syncthreads();

    // ================= stem: im2col GEMM -> PReLU -> LDS image =================
    const float* Ximg = X + (size_t)img * 3 * HI * HI;
#pragma unroll 1
    for (int t = 0; t < MTIW; ++t) {
        if (wave + NW * t >= MTI) break;      // wave-uniform
        const int p = 32 * (wave + NW * t) + r;
        const int iy = p / W, ix = p - iy * W;
        const int gy = row0 + iy;
        const bool ok = (p < PIN) && (gy >= 0) && (gy < H);
        const int base = (2 * gy - 1) * HI + (2 * ix - 1);
        unsigned mask = ok ? live : 0u;
        if (gy == 0) mask &= ~ky0;            // input row -1 is padding
        if (ix == 0) mask &= ~kx0;            // input column -1 is padding
        V8 bfr[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool v = (mask >> (8 * s + j)) & 1u;
                const float f = v ? Ximg[base + toff[s][j]] : 0.0f;
                bfr[s][j] = (T)f;
            }
        const int slot = iy * PITCH + ix + 1;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(lds + OFF_B1 + (32 * m + 16 * (q >> 1) + 8 * h + 4 * (q & 1)) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[4 * q + i] = b[i];
            }
            acc = mfma32(wa[m][0], bfr[0], acc);
            acc = mfma32(wa[m][1], bfr[1], acc);
            if (ok) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    V8 o;
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const f32x4 sl = *reinterpret_cast<const f32x4*>(lds + OFF_S1 + (32 * m + 16 * g + 8 * h + 4 * a) * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v = acc[4 * (2 * g + a) + i];
                            o[4 * a + i] = (T)fmaf(fminf(v, 0.0f), sl[i], v);   // PReLU: sl = slope - 1
                        }
                    }
                    const int u = 4 * m + 2 * g + h;       // channels [8u, 8u+8)
                    *reinterpret_cast<V8*>(es + slot * 128 + ((u ^ eswz2(slot)) << 4)) = o;
                }
            }
        }
    }
    __syncthreads();

    // ================= depthwise 3x3 + PReLU from the LDS image -> NHWC =================
    T* Oimg = O + (size_t)img * H * W * 64;
#pragma unroll 1
    for (int t = 0; t < MTOW; ++t) {
        if (wave + NW * t >= MTO) break;      // wave-uniform
        const int po = 32 * (wave + NW * t) + r;
        const bool ok = po < POUT;
        const int pc = ok ? po : 0;
        const int oy = pc / W, ox = pc - oy * W;
        int tap_off[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int slot = (oy + ky) * PITCH + ox + kx;
                tap_off[ky * 3 + kx] = slot * 128 + (eswz2(slot) << 4);
            }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int u = 2 * s4 + h;
            const f32x4 b2a = *reinterpret_cast<const f32x4*>(lds + OFF_B2 + u * 32),
                        b2b = *reinterpret_cast<const f32x4*>(lds + OFF_B2 + u * 32 + 16);
            float d[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { d[j] = b2a[j]; d[4 + j] = b2b[j]; }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const V8 e = *reinterpret_cast<const V8*>(es + (tap_off[ky * 3 + kx] ^ (u << 4)));
                    const V8 wk = *reinterpret_cast<const V8*>(lds + OFF_W2 + ((ky * 3 + kx) * 64 + 8 * u) * 2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[j] = fmaf((float)e[j], (float)wk[j], d[j]);
                }
            const f32x4 s2a = *reinterpret_cast<const f32x4*>(lds + OFF_S2 + u * 32),
                        s2b = *reinterpret_cast<const f32x4*>(lds + OFF_S2 + u * 32 + 16);
            V8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (T)fmaf(fminf(d[j], 0.0f), s2a[j], d[j]);                  // PReLU: s2 = slope - 1
                o[4 + j] = (T)fmaf(fminf(d[4 + j], 0.0f), s2b[j], d[4 + j]);
            }
            if (ok) *reinterpret_cast<V8*>(Oimg + ((size_t)(y0 + oy) * W + ox) * 64 + 8 * u) = o;
        }
    }
}

template <typename T, int TR, int NW, int MINW>
static int launch_stemdw_cfg(const StemDwParams& p, const float* x, void* out, int B, hipStream_t s) {
    constexpr int lds = (TR + 2) * 58 * 128 + 9 * 64 * 2 + 4 * 256;
    auto kern = stemdw_kernel<T, TR, NW, MINW>;
    static bool attr_done = false;
    if (!attr_done) {
        FRA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(56 / TR, B), dim3(NW * 64), lds, s, x, (const T*)p.wa, p.b1, p.s1, (const T*)p.w2, p.b2,
                       p.s2, (T*)out);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

int launch_stemdw(int dt, const StemDwParams& p, const float* x, void* out, int B, hipStream_t s) {
    FRA_REQUIRE(x && out && p.wa && p.b1 && p.s1 && p.w2 && p.b2 && p.s2, FRA_ERR_ARG, "stemdw: null pointer");
    FRA_REQUIRE(B > 0 && B <= 65535, FRA_ERR_ARG, "stemdw: batch %d not in [1,65535]", B);
    if (dt == ACT_F16) return launch_stemdw_cfg<f16, 4, 4, 3>(p, x, out, B, s);
    if (dt == ACT_BF16) return launch_stemdw_cfg<bf16, 4, 4, 3>(p, x, out, B, s);
    FRA_REQUIRE(false, FRA_ERR_ARG, "stemdw: unsupported activation dtype %d", dt);
    return FRA_ERR_ARG;
}

}  // namespace fra
