// What this GPU sustains, to put beside the datasheet peaks the rooflines divide by: a streaming copy
// (16 bytes per lane, grid-stride), a read-only pass, and register-resident MFMA loops (f16 32x32x16 and 16x16x32;
// inline asm with VGPR accumulators, dependent MFMAs 4 resp. 8 instructions apart, so no wait states are needed).
// Built by scripts/diag/build_peaks.sh into scripts/diag/libpeaks.so; run by scripts/diag/peaks.py.  Not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_kernel(const f4v* __restrict__ src, f4v* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

__global__ __launch_bounds__(256) void read_kernel(const f4v* __restrict__ src, float* __restrict__ sink, size_t n) {
    f4v a = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const f4v v = src[i];
        a += v;
    }
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) sink[0] = a[0];   // never true for the zero-filled source; keeps the loads
}

template <int KIND>
__global__ __launch_bounds__(256) void mfma_kernel(float* __restrict__ sink, int iters) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (j + 1)); }
    float s = 0.f;
    if (KIND == 0) {
        f16v c[4];
        for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) c[k][j] = 0.f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[k]) : "v"(a), "v"(b));
        for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) s += c[k][j];
    } else {
        f4v c[8];
        for (int k = 0; k < 8; ++k) for (int j = 0; j < 4; ++j) c[k][j] = 0.f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[k]) : "v"(a), "v"(b));
        for (int k = 0; k < 8; ++k) for (int j = 0; j < 4; ++j) s += c[k][j];
    }
    if (s == 12345.678f) sink[0] = s;
}

static float timed(hipStream_t st, int reps, void (*launch)(hipStream_t, void*), void* ctx) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(st, ctx);                       // warm-up
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) launch(st, ctx);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / reps;
}

struct Ctx { const void* src; void* dst; float* sink; size_t n; int iters; int blocks; };

extern "C" {
// bytes: size of each buffer (multiple of 16); returns ms per pass, or -1 on a launch error
float peaks_copy_ms(const void* src, void* dst, size_t bytes, int reps) {
    Ctx c{src, dst, nullptr, bytes / 16, 0, 256 * 32};
    const float ms = timed(0, reps, [](hipStream_t st, void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(copy_kernel, dim3(c->blocks), dim3(256), 0, st, (const f4v*)c->src, (f4v*)c->dst, c->n); }, &c);
    return hipGetLastError() == hipSuccess ? ms : -1.f;
}
float peaks_read_ms(const void* src, float* sink, size_t bytes, int reps) {
    Ctx c{src, nullptr, sink, bytes / 16, 0, 256 * 32};
    const float ms = timed(0, reps, [](hipStream_t st, void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(read_kernel, dim3(c->blocks), dim3(256), 0, st, (const f4v*)c->src, c->sink, c->n); }, &c);
    return hipGetLastError() == hipSuccess ? ms : -1.f;
}
// kind 0: 32x32x16, kind 1: 16x16x32; blocks of 4 waves; returns ms per launch.  FLOP per launch = blocks * 4 waves * iters * 131072 (both kinds)
float peaks_mfma_ms(int kind, float* sink, int blocks, int iters, int reps) {
    Ctx c{nullptr, nullptr, sink, 0, iters, blocks};
    float ms;
    if (kind == 0) ms = timed(0, reps, [](hipStream_t st, void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(mfma_kernel<0>, dim3(c->blocks), dim3(256), 0, st, c->sink, c->iters); }, &c);
    else ms = timed(0, reps, [](hipStream_t st, void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(mfma_kernel<1>, dim3(c->blocks), dim3(256), 0, st, c->sink, c->iters); }, &c);
    return hipGetLastError() == hipSuccess ? ms : -1.f;
}
}
