// Operand layout and issue rate of v_smfmac_f32_16x16x64_f16 on gfx950.
//   probe <out.bin>   writes A values [64][8] f16, idx [64] u32, B [64][16] f16, D [64][4] f32 of one instruction on
//                     random operands, then prints cycles per instruction of dependent chains (sparse and dense).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void one(const h8* a, const h16* b, const int* idx, f4* d) {
    const int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a[l], b[l], c, idx[l], 0, 0);
    d[l] = c;
}

template <int SPARSE, int CHAINS>
__global__ void rate(const h8* a, const h16* b, const int* idx, f4* d, int iters, long long* cycles) {
    const int l = threadIdx.x & 63;
    h8 av = a[l];
    h16 bv = b[l];
    h8 bd = {bv[0], bv[1], bv[2], bv[3], bv[4], bv[5], bv[6], bv[7]};
    const int ix = idx[l];
    f4 c[CHAINS];
    for (int i = 0; i < CHAINS; ++i) c[i] = f4{0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (SPARSE) c[i] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(av, bv, c[i], ix, 0, 0);
            else c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bd, c[i], 0, 0, 0);
        }
    const long long t1 = __builtin_readcyclecounter();
    f4 s = c[0];
    for (int i = 1; i < CHAINS; ++i) s += c[i];
    d[threadIdx.x] = s;
    if (threadIdx.x == 0) *cycles = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    std::vector<_Float16> A(64 * 8), B(64 * 16);
    std::vector<unsigned> I(64);
    std::vector<float> D(64 * 4);
    srand(7);
    for (auto& v : A) v = (_Float16)((rand() % 15) - 7);
    for (auto& v : B) v = (_Float16)((rand() % 9) - 4);
    for (auto& v : I) v = (unsigned)rand() ^ ((unsigned)rand() << 16);
    void *a, *b, *ix, *d; long long* cyc;
    CK(hipMalloc(&a, A.size() * 2)); CK(hipMalloc(&b, B.size() * 2)); CK(hipMalloc(&ix, 256)); CK(hipMalloc(&d, 256 * 16));
    CK(hipMalloc((void**)&cyc, 8));
    CK(hipMemcpy(a, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(ix, I.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, (const h8*)a, (const h16*)b, (const int*)ix, (f4*)d);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), d, 64 * 16, hipMemcpyDeviceToHost));
    if (argc > 1) {
        FILE* f = fopen(argv[1], "wb");
        fwrite(A.data(), 2, A.size(), f); fwrite(I.data(), 4, 64, f); fwrite(B.data(), 2, B.size(), f); fwrite(D.data(), 4, D.size(), f);
        fclose(f);
    }
    const int iters = 20000;
    long long c = 0;
#define RATE(SP, CH, WAVES) \
    hipLaunchKernelGGL((rate<SP, CH>), dim3(1), dim3(64 * WAVES), 0, 0, (const h8*)a, (const h16*)b, (const int*)ix, (f4*)d, iters, cyc); \
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); \
    printf("%s chains %d waves/SIMD %d: %.2f cycles (readcyclecounter ticks) per instruction per wave\n", SP ? "smfmac 16x16x64" : "mfma 16x16x32 ", CH, WAVES / 4 ? WAVES / 4 : 1, (double)c / iters / CH);
    RATE(0, 1, 1) RATE(1, 1, 1) RATE(0, 4, 1) RATE(1, 4, 1) RATE(0, 4, 8) RATE(1, 4, 8)
    return 0;
}
