// What a chain of short dependent kernels costs (a) in stream order on one stream and (b) on two streams in turn, each
// kernel waiting for its predecessor through a flag in global memory after it has done the part of its work that does
// not depend on it (its "weights"): eager launches and a replayed graph.
//   hipcc --offload-arch=gfx950 -O2 probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void spin_cycles(long long n) {
    const long long t0 = clock64();
    while (clock64() - t0 < n) __builtin_amdgcn_s_sleep(1);
}

// weights: 64 KB per workgroup read first; in / out: 8 KB per workgroup; wait / done / reset: flags (may be null)
__global__ __launch_bounds__(512) void work(const float4* __restrict__ weights, const float4* __restrict__ in, float4* __restrict__ out,
                                            int* wait, int* done, int* reset, long long work_cycles, unsigned* fault) {
    __shared__ float4 sh[512];
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) {                          // the part that does not depend on the predecessor
        const float4 w = weights[(blockIdx.x * 8 + i) * 512 + threadIdx.x];
        acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (wait) {
        if (threadIdx.x < 64) {
            const int nb = (int)blockIdx.x + (int)(threadIdx.x % 3) - 1;        // the predecessor's workgroups next to this one
            const bool valid = nb >= 0 && nb < (int)gridDim.x;
            int spins = 0;
            for (;;) {
                const int v = valid ? __hip_atomic_load(wait + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1;
                if (__builtin_amdgcn_ballot_w64(v == 0) == 0) break;
                if (++spins > (1 << 20)) { if (threadIdx.x == 0) *fault = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    }
    if (reset && threadIdx.x == 0) __hip_atomic_store(reset + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float4 x = in[blockIdx.x * 512 + threadIdx.x];
    spin_cycles(work_cycles);
    float4 o = {x.x + acc.x + 1.f, x.y + sh[(threadIdx.x + 1) & 511].y, x.z, x.w};
    out[blockIdx.x * 512 + threadIdx.x] = o;
    if (done) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done + blockIdx.x, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    const int NK = 6, WG = 7;
    float4 *w, *buf[2]; int* flags; unsigned* fault;
    CK(hipMalloc(&w, (size_t)NK * WG * 8 * 512 * 16)); CK(hipMemset(w, 0, (size_t)NK * WG * 8 * 512 * 16));
    for (auto& b : buf) { CK(hipMalloc(&b, WG * 512 * 16)); CK(hipMemset(b, 0, WG * 512 * 16)); }
    CK(hipMalloc(&flags, NK * 64 * 4)); CK(hipMemset(flags, 0, NK * 64 * 4));
    CK(hipMalloc(&fault, 4)); CK(hipMemset(fault, 0, 4));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    for (long long cycles : {2000LL, 12000LL}) {
        auto chain = [&](bool flagged) -> int {
            if (flagged) { CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); }
            for (int k = 0; k < NK; ++k) {
                hipStream_t s = flagged && (k & 1) ? s2 : s1;
                int* wait = flagged && k > 0 ? flags + 64 * (k - 1) : nullptr;
                int* done = flagged ? flags + 64 * k : nullptr;
                int* reset = !flagged ? nullptr : (k >= 2 ? flags + 64 * (k - 2) : flags + 64 * (NK - 2 + k));
                hipLaunchKernelGGL(work, dim3(WG), dim3(512), 0, s, w + (size_t)k * WG * 8 * 512, buf[k & 1], buf[(k + 1) & 1], wait, done, reset, cycles, fault);
            }
            if (flagged) { CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); }
            return 0;
        };
        for (int flagged = 0; flagged < 2; ++flagged) {
            // eager
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipDeviceSynchronize());
                const int N = 300;
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; ++i) { if (chain(flagged)) return 1; CK(hipStreamSynchronize(s1)); }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
                if (rep) printf("work %5lld cycles, %s, eager : %.1f us per chain of %d kernels (%.2f per kernel)\n", cycles, flagged ? "two streams + flags" : "one stream         ", us, NK, us / NK);
            }
            // graph
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
            if (chain(flagged)) return 1;
            CK(hipStreamEndCapture(s1, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipDeviceSynchronize());
                const int N = 300;
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; ++i) { CK(hipGraphLaunch(ge, s1)); CK(hipStreamSynchronize(s1)); }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
                if (rep) printf("work %5lld cycles, %s, graph : %.1f us per chain of %d kernels (%.2f per kernel)\n", cycles, flagged ? "two streams + flags" : "one stream         ", us, NK, us / NK);
            }
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    unsigned f = 0; CK(hipMemcpy(&f, fault, 4, hipMemcpyDeviceToHost));
    printf("waits that gave up: %u\n", f);
    return 0;
}
