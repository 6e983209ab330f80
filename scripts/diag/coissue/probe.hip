// Do a SIMD's two waves overlap matrix and vector work?  One workgroup of 8 waves (waves w and w + 4 share a SIMD):
// waves 0-3 run role A, waves 4-7 role B, each a loop of independent instructions; prints cycles per loop for each
// role alone and for the pair.  Roles: 0 idle, 1 MFMA 16x16x32 f16 (4 chains), 2 VALU fma (8 chains), 3 LDS ds_read_b128.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int ROLE> __device__ void work(int iters, float* out, const h8* a, char* lds) {
    const int l = threadIdx.x & 63;
    if (ROLE == 1) {
        h8 av = a[l], bv = a[64 + l];
        f4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c[i], 0, 0, 0);
        out[threadIdx.x] = c[0][0] + c[1][0] + c[2][0] + c[3][0];
    } else if (ROLE == 2) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = l + i;
        const float m = out[0];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(m));
        float s = 0;
        for (int i = 0; i < 8; ++i) s += v[i];
        out[threadIdx.x] = s;
    } else if (ROLE == 3) {
        f4 r[4];
        const int addr = l * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(i * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
        }
        out[threadIdx.x] = r[0][0] + r[1][0] + r[2][0] + r[3][0];
    }
}

template <int RA, int RB> __global__ void k(int iters, float* out, const h8* a, long long* cyc) {
    __shared__ char lds[8192];
    if (threadIdx.x < 2048) reinterpret_cast<float*>(lds)[threadIdx.x] = 1.0f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const long long t0 = __builtin_readcyclecounter();
    if (wave < 4) work<RA>(iters, out, a, lds);
    else work<RB>(iters, out, a, lds);
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int RA, int RB> int run(const char* name, float* out, h8* a, long long* cyc, int per_a, int per_b) {
    const int iters = 20000;
    hipLaunchKernelGGL((k<RA, RB>), dim3(1), dim3(512), 0, 0, iters, out, (const h8*)a, cyc);
    CK(hipDeviceSynchronize());
    long long c[8];
    CK(hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost));
    printf("%-28s waves 0-3: %7.2f ticks per instruction   waves 4-7: %7.2f\n", name,
           per_a ? (double)c[0] / iters / per_a : 0.0, per_b ? (double)c[4] / iters / per_b : 0.0);
    return 0;
}
int main() {
    float* out; h8* a; long long* cyc;
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&a, 4096)); CK(hipMalloc(&cyc, 64));
    CK(hipMemset(out, 0, 4096)); CK(hipMemset(a, 0, 4096));
    run<1, 0>("mfma alone", out, a, cyc, 4, 0);
    run<2, 0>("valu alone", out, a, cyc, 8, 0);
    run<3, 0>("lds alone", out, a, cyc, 4, 0);
    run<1, 1>("mfma + mfma", out, a, cyc, 4, 4);
    run<2, 2>("valu + valu", out, a, cyc, 8, 8);
    run<1, 2>("mfma + valu", out, a, cyc, 4, 8);
    run<1, 3>("mfma + lds", out, a, cyc, 4, 4);
    run<2, 3>("valu + lds", out, a, cyc, 8, 4);
    return 0;
}
