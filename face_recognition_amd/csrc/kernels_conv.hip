// Direct convolution kernels for gfx950: the dense 3x3 stem, depthwise 3x3 and the 7x7 global
// depthwise conv, plus the small row-normalisation kernels.  Activations are NHWC in a 16-bit
// type T (f16 or bf16); every thread moves 8 channels = 16 bytes per access so that a wave's
// accesses cover whole 128-byte lines.  BatchNorm is already folded into weights and bias; the
// per-channel PReLU runs in the epilogue.  All arithmetic accumulates in fp32.
#include "common.h"

namespace fra {

template <typename T> __device__ __forceinline__ float tof(T v) { return (float)v; }

// ---------------------------------------------------------------------------------------------
// Stem: Conv2d(3,64,3,stride 2,pad 1) + bias + PReLU.   (reference: face_model.py:80, 26-36)
// One thread per output pixel; the 27 input taps sit in registers, the weights are wave-uniform
// and come through scalar loads.  Input is the caller's fp32 NCHW tensor, output NHWC T.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ slope, T* __restrict__ out,
                                                   int total /* B*56*56 */) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int b = idx / 3136, p = idx - b * 3136;
    const int yo = p / 56, xo = p - yo * 56;
    float xin[27];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* plane = x + ((size_t)b * 3 + c) * (112 * 112);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * yo - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * xo - 1 + kx;
                const bool ok = (iy >= 0) && (ix >= 0);  // iy, ix <= 111 always holds for stride 2, pad 1
                xin[(c * 3 + ky) * 3 + kx] = ok ? plane[iy * 112 + ix] : 0.0f;
            }
        }
    }
    typedef typename vec8<T>::type V8;
    V8* orow = reinterpret_cast<V8*>(out + (size_t)idx * 64);
#pragma unroll 1
    for (int ocb = 0; ocb < 8; ++ocb) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bias[ocb * 8 + j];
#pragma unroll
        for (int t = 0; t < 27; ++t) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(xin[t], w[t * 64 + ocb * 8 + j], acc[j]);
        }
        V8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = acc[j];
            o[j] = (T)(v > 0.0f ? v : v * slope[ocb * 8 + j]);
        }
        orow[ocb] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// Depthwise 3x3, pad 1, stride 1 or 2, + bias + optional PReLU.   (face_model.py:52, 81)
// A thread owns 8 channels of one output column and walks a strip of 7 output rows, keeping a
// sliding 3-row window of the input in registers, so each input row is read once per strip.
// Lanes run fastest over channels, then over x: a wave reads contiguous NHWC memory.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Px3 { typename vec8<T>::type v[3]; };

template <typename T>
__device__ __forceinline__ Px3<T> load_row3(const T* __restrict__ img, int y, int xc, int H, int W, int C, int c0,
                                            int xstep) {
    // pixels (y, xc-1), (y, xc), (y, xc+1) of one image; zero outside the image
    typedef typename vec8<T>::type V8;
    Px3<T> r;
    const bool yok = (y >= 0) && (y < H);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int x = xc - 1 + j;
        V8 z = {};
        if (yok && x >= 0 && x < W) z = *reinterpret_cast<const V8*>(img + ((size_t)y * W + x) * C + c0);
        r.v[j] = z;
    }
    (void)xstep;
    return r;
}

template <typename T, int STRIDE>
__global__ __launch_bounds__(256) void dw3x3_kernel(const T* __restrict__ in, const T* __restrict__ w,
                                                    const float* __restrict__ bias,
                                                    const float* __restrict__ slope, T* __restrict__ out, int H,
                                                    int W, int C, int Ho, int Wo, long total_threads) {
    typedef typename vec8<T>::type V8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_threads) return;
    const int CG = C >> 3;
    const int strips = Ho / 7;
    const int cg = (int)(i % CG);
    long rest = i / CG;
    const int xo = (int)(rest % Wo);
    rest /= Wo;
    const int strip = (int)(rest % strips);
    const long b = rest / strips;
    const int c0 = cg * 8;
    const int y0 = strip * 7;

    V8 wv[9];  // kept packed: the mixed-precision FMA reads 16-bit operands directly
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const V8*>(w + (size_t)k * C + c0);
    float bi[8], sl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bi[j] = bias[c0 + j];
        sl[j] = slope ? slope[c0 + j] : 1.0f;
    }
    const T* img = in + (size_t)b * H * W * C;
    T* oimg = out + (size_t)b * Ho * Wo * C;
    const int xc = xo * STRIDE;  // centre column in the input

    Px3<T> r0, r1, r2;
    if (STRIDE == 1) {
        r0 = load_row3<T>(img, y0 - 1, xc, H, W, C, c0, 1);
        r1 = load_row3<T>(img, y0, xc, H, W, C, c0, 1);
    } else {
        r0 = load_row3<T>(img, 2 * y0 - 1, xc, H, W, C, c0, 1);
    }
#pragma unroll 1
    for (int oy = 0; oy < 7; ++oy) {
        const int yo = y0 + oy;
        if (STRIDE == 1) {
            r2 = load_row3<T>(img, yo + 1, xc, H, W, C, c0, 1);
        } else {
            r1 = load_row3<T>(img, 2 * yo, xc, H, W, C, c0, 1);
            r2 = load_row3<T>(img, 2 * yo + 1, xc, H, W, C, c0, 1);
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bi[j];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] = fmaf(tof(r0.v[kx][j]), tof(wv[0 + kx][j]), acc[j]);
                acc[j] = fmaf(tof(r1.v[kx][j]), tof(wv[3 + kx][j]), acc[j]);
                acc[j] = fmaf(tof(r2.v[kx][j]), tof(wv[6 + kx][j]), acc[j]);
            }
        }
        V8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = acc[j];
            o[j] = (T)(v > 0.0f ? v : v * sl[j]);
        }
        *reinterpret_cast<V8*>(oimg + ((size_t)yo * Wo + xo) * C + c0) = o;
        if (STRIDE == 1) {
            r0 = r1;
            r1 = r2;
        } else {
            r0 = r2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Global depthwise conv 7x7 on a 7x7 map (output 1x1) + bias, no activation.  (face_model.py:89)
// One thread per (image, 8 channels); 49 independent 16-byte loads in flight.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gdconv7_kernel(const T* __restrict__ in, const T* __restrict__ w,
                                                      const float* __restrict__ bias, T* __restrict__ out, int C,
                                                      long total_threads) {
    typedef typename vec8<T>::type V8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_threads) return;
    const int CG = C >> 3;
    const int cg = (int)(i % CG);
    const long b = i / CG;
    const int c0 = cg * 8;
    const T* img = in + (size_t)b * 49 * C + c0;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = bias[c0 + j];
#pragma unroll 7
    for (int t = 0; t < 49; ++t) {
        const V8 xv = *reinterpret_cast<const V8*>(img + (size_t)t * C);
        const V8 wv = *reinterpret_cast<const V8*>(w + (size_t)t * C + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(tof(xv[j]), tof(wv[j]), acc[j]);
    }
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (T)acc[j];
    *reinterpret_cast<V8*>(out + (size_t)b * C + c0) = o;
}

// ---------------------------------------------------------------------------------------------
// out = (a + b) / ||a + b||_2 per row (b optional).  One wave per row.  No epsilon, as in
// face_model.py:19-22: an all-zero row yields NaN.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_l2norm_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, int B, int E) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* pa = a + (size_t)row * E;
    const float* pb = b ? b + (size_t)row * E : nullptr;
    float ss = 0.0f;
    for (int e = lane; e < E; e += 64) {
        const float v = pa[e] + (pb ? pb[e] : 0.0f);
        ss = fmaf(v, v, ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float nrm = sqrtf(ss);
    for (int e = lane; e < E; e += 64) {
        const float v = pa[e] + (pb ? pb[e] : 0.0f);
        out[(size_t)row * E + e] = v / nrm;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = tof(in[i]);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline unsigned blocks_for(long threads) { return (unsigned)((threads + 255) / 256); }

int launch_stem(int dt, const float* x, const float* w, const float* bias, const float* slope, void* out, int B,
                hipStream_t s) {
    FRA_REQUIRE(x && w && bias && slope && out, FRA_ERR_ARG, "stem: null pointer");
    FRA_REQUIRE(B > 0 && (long)B * 3136 < 2147483647L, FRA_ERR_ARG, "stem: batch %d out of range", B);
    const int total = B * 3136;
    if (dt == ACT_F16)
        hipLaunchKernelGGL(stem_kernel<f16>, dim3(blocks_for(total)), dim3(256), 0, s, x, w, bias, slope, (f16*)out,
                           total);
    else if (dt == ACT_BF16)
        hipLaunchKernelGGL(stem_kernel<bf16>, dim3(blocks_for(total)), dim3(256), 0, s, x, w, bias, slope,
                           (bf16*)out, total);
    else
        FRA_REQUIRE(false, FRA_ERR_ARG, "stem: unsupported activation dtype %d", dt);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

template <typename T>
static int launch_dw3x3_t(const T* in, const T* w, const float* bias, const float* slope, T* out, int B, int H,
                          int W, int C, int stride, hipStream_t s) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    FRA_REQUIRE(Ho % 7 == 0, FRA_ERR_ARG, "dw3x3: output height %d is not a multiple of 7", Ho);
    const long threads = (long)B * (Ho / 7) * Wo * (C / 8);
    FRA_REQUIRE(threads / 256 < 2147483647L, FRA_ERR_ARG, "dw3x3: problem too large");
    if (stride == 1)
        hipLaunchKernelGGL((dw3x3_kernel<T, 1>), dim3(blocks_for(threads)), dim3(256), 0, s, in, w, bias, slope, out,
                           H, W, C, Ho, Wo, threads);
    else
        hipLaunchKernelGGL((dw3x3_kernel<T, 2>), dim3(blocks_for(threads)), dim3(256), 0, s, in, w, bias, slope, out,
                           H, W, C, Ho, Wo, threads);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

int launch_dw3x3(int dt, const void* in, const void* w, const float* bias, const float* slope, void* out, int B,
                 int H, int W, int C, int stride, hipStream_t s) {
    FRA_REQUIRE(in && w && bias && out, FRA_ERR_ARG, "dw3x3: null pointer");
    FRA_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, FRA_ERR_ARG, "dw3x3: bad shape B=%d H=%d W=%d C=%d",
                B, H, W, C);
    FRA_REQUIRE(stride == 1 || stride == 2, FRA_ERR_ARG, "dw3x3: stride %d unsupported", stride);
    if (dt == ACT_F16)
        return launch_dw3x3_t<f16>((const f16*)in, (const f16*)w, bias, slope, (f16*)out, B, H, W, C, stride, s);
    if (dt == ACT_BF16)
        return launch_dw3x3_t<bf16>((const bf16*)in, (const bf16*)w, bias, slope, (bf16*)out, B, H, W, C, stride, s);
    FRA_REQUIRE(false, FRA_ERR_ARG, "dw3x3: unsupported activation dtype %d", dt);
}

int launch_gdconv7(int dt, const void* in, const void* w, const float* bias, void* out, int B, int C,
                   hipStream_t s) {
    FRA_REQUIRE(in && w && bias && out, FRA_ERR_ARG, "gdconv7: null pointer");
    FRA_REQUIRE(B > 0 && C > 0 && C % 8 == 0, FRA_ERR_ARG, "gdconv7: bad shape B=%d C=%d", B, C);
    const long threads = (long)B * (C / 8);
    if (dt == ACT_F16)
        hipLaunchKernelGGL(gdconv7_kernel<f16>, dim3(blocks_for(threads)), dim3(256), 0, s, (const f16*)in,
                           (const f16*)w, bias, (f16*)out, C, threads);
    else if (dt == ACT_BF16)
        hipLaunchKernelGGL(gdconv7_kernel<bf16>, dim3(blocks_for(threads)), dim3(256), 0, s, (const bf16*)in,
                           (const bf16*)w, bias, (bf16*)out, C, threads);
    else
        FRA_REQUIRE(false, FRA_ERR_ARG, "gdconv7: unsupported activation dtype %d", dt);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

int launch_add_l2norm(const float* a, const float* b, float* out, int B, int E, hipStream_t s) {
    FRA_REQUIRE(a && out, FRA_ERR_ARG, "l2norm: null pointer");
    FRA_REQUIRE(B >= 0 && E > 0, FRA_ERR_ARG, "l2norm: bad shape B=%d E=%d", B, E);
    if (B == 0) return FRA_OK;
    hipLaunchKernelGGL(add_l2norm_kernel, dim3((B + 3) / 4), dim3(256), 0, s, a, b, out, B, E);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

int launch_to_f32(int dt, const void* in, float* out, int64_t n, hipStream_t s) {
    FRA_REQUIRE(in && out && n > 0, FRA_ERR_ARG, "to_f32: bad argument");
    if (dt == ACT_F16)
        hipLaunchKernelGGL(to_f32_kernel<f16>, dim3(blocks_for(n)), dim3(256), 0, s, (const f16*)in, out, (long)n);
    else if (dt == ACT_BF16)
        hipLaunchKernelGGL(to_f32_kernel<bf16>, dim3(blocks_for(n)), dim3(256), 0, s, (const bf16*)in, out, (long)n);
    else
        FRA_REQUIRE(false, FRA_ERR_ARG, "to_f32: unsupported dtype %d", dt);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

}  // namespace fra
