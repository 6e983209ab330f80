// What a cooperative launch costs between dependent kernels on one stream, and whether a stream capture takes it:
//   hipcc --offload-arch=gfx950 -O2 probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int main() {
    int* d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    void* args[] = {&d};
    const int N = 500;
    for (int mode = 0; mode < 3; ++mode) {
        // 0: ordinary launches, 1: cooperative launches, 2: five ordinary launches then one cooperative, repeated
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                const bool coop = mode == 1 || (mode == 2 && i % 6 == 5);
                if (coop) CK(hipLaunchCooperativeKernel((const void*)tiny, dim3(7), dim3(512), args, 0, s));
                else hipLaunchKernelGGL(tiny, dim3(7), dim3(512), 0, s, d);
            }
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep) printf("mode %d: %.2f us per dependent launch\n", mode, us);
        }
    }
    // stream capture with a cooperative launch inside
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(tiny, dim3(7), dim3(512), 0, s, d);
    hipError_t e = hipLaunchCooperativeKernel((const void*)tiny, dim3(7), dim3(512), args, 0, s);
    printf("cooperative launch under capture: %s\n", hipGetErrorString(e));
    hipLaunchKernelGGL(tiny, dim3(7), dim3(512), 0, s, d);
    e = hipStreamEndCapture(s, &g);
    printf("end capture: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        printf("instantiate: %s\n", hipGetErrorString(e));
        if (e == hipSuccess) {
            CK(hipMemset(d, 0, 4));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipStreamSynchronize(s));
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 200; ++i) CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
                if (rep) printf("graph replay (3 kernels, the middle one cooperative): %.2f us\n", us);
            }
            int h = 0; CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
            printf("kernels run by 400 replays: %d (1200 expected)\n", h);
        }
    }
    return 0;
}
