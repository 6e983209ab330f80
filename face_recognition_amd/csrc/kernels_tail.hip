// Fused network tail for gfx950: conv_6_sep (1x1, 128 -> 512, PReLU) + conv_6_dw (7x7 global depthwise,
// 512 channels, linear) in one kernel (reference: face_model.py:88-89).  The 7x7x512 tensor between the two
// never exists in memory: it is 25 KB per face against 12.5 KB of input and 1 KB of output, and writing and
// re-reading it was all the two separate kernels did with their time.
//
// A workgroup of 8 waves takes TAIL_IMG faces.  Their 49 x 128 inputs are staged in LDS (rows padded by one
// 16-byte unit, rows 49..63 zero).  A wave owns two 32-channel tiles of the 512 outputs; for a tile it keeps
// the 1x1 weights as eight B fragments in registers (k = 128) and the 7x7 weights of its channel in 32
// registers, then for every face and both 32-pixel tiles runs eight 32x32x16 MFMAs with pixels as rows and
// channels as columns: a lane then holds ONE channel and 16 pixels, so bias, PReLU and the depthwise sum over
// pixels are 16 in-lane operations, and one cross-half add finishes the 49-pixel dot product.
#include "common.h"

namespace fra {

constexpr int TAIL_IMG = 4, TAIL_NW = 8, TAIL_K = 128, TAIL_C = 512, TAIL_PX = 49;
constexpr int TAIL_PITCH = TAIL_K * 2 + 16;                     // LDS bytes per pixel row
constexpr int TAIL_LDS = TAIL_IMG * 64 * TAIL_PITCH;

// x T [B][49][128] (NHWC); w6 T [512][128]; b6, s6 f32 [512] (s6 = PReLU slope - 1); w7 T [49][512]; b7 f32 [512];
// out T [B][512]
template <typename T>
__global__ __launch_bounds__(TAIL_NW * 64) void tail_kernel(const T* __restrict__ x, const T* __restrict__ w6,
                                                            const float* __restrict__ b6, const float* __restrict__ s6,
                                                            const T* __restrict__ w7, const float* __restrict__ b7,
                                                            T* __restrict__ out, int B) {
    typedef typename vec8<T>::type V8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int img0 = blockIdx.x * TAIL_IMG;

    // ---- stage the inputs: 16 units per pixel row, zero rows for pixels 49..63 and for faces past the batch ----
    for (int i = tid; i < TAIL_IMG * 64 * 16; i += TAIL_NW * 64) {
        const int u = i & 15, px = (i >> 4) & 63, im = i >> 10;
        V8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (T)0.0f;
        if (px < TAIL_PX && img0 + im < B) v = *reinterpret_cast<const V8*>(x + ((size_t)(img0 + im) * TAIL_PX + px) * TAIL_K + u * 8);
        *reinterpret_cast<V8*>(lds + (im * 64 + px) * TAIL_PITCH + u * 16) = v;
    }
    __syncthreads();

    // gridDim.y workgroups share the channel tiles of a group of faces (small batches: more compute units at work)
#pragma unroll 1
    for (int ct = wave + TAIL_NW * blockIdx.y; ct < TAIL_C / 32; ct += TAIL_NW * gridDim.y) {
        const int ch = 32 * ct + r;                              // this lane's output channel
        V8 wb[TAIL_K / 16];
#pragma unroll
        for (int ks = 0; ks < TAIL_K / 16; ++ks) wb[ks] = *reinterpret_cast<const V8*>(w6 + (size_t)ch * TAIL_K + 16 * ks + 8 * h);
        const float bias = b6[ch], slope = s6[ch];
        // 7x7 weights of this channel for the pixels this lane's accumulator registers hold (0 past pixel 48)
        float wv[2][16];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int px = 32 * pt + acc_row(g, h);
                wv[pt][g] = px < TAIL_PX ? (float)w7[(size_t)px * TAIL_C + ch] : 0.0f;
            }
#pragma unroll 1
        for (int im = 0; im < TAIL_IMG; ++im) {
            if (img0 + im >= B) break;                           // uniform
            float sum = 0.0f;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = bias;
                const char* row = lds + (im * 64 + 32 * pt + r) * TAIL_PITCH + 16 * h;
#pragma unroll
                for (int ks = 0; ks < TAIL_K / 16; ++ks)
                    acc = mfma32(*reinterpret_cast<const V8*>(row + 32 * ks), wb[ks], acc);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float v = acc[g];
                    sum = fmaf(fmaf(neg_part(v), slope, v), wv[pt][g], sum);
                }
            }
            sum += __shfl_xor(sum, 32, 64);                      // the other half holds the other 16 + 16 pixels
            if (h == 0) out[(size_t)(img0 + im) * TAIL_C + ch] = (T)(sum + b7[ch]);
        }
    }
}

}  // namespace fra

using namespace fra;

int fra::launch_tail(int dt, const void* x, const void* w6, const float* b6, const float* s6, const void* w7,
                     const float* b7, void* out, int B, hipStream_t s) {
    FRA_REQUIRE(x && w6 && b6 && s6 && w7 && b7 && out, FRA_ERR_ARG, "tail: null pointer");
    FRA_REQUIRE(B > 0, FRA_ERR_ARG, "tail: batch %d", B);
    const int groups = (B + TAIL_IMG - 1) / TAIL_IMG;
    // a workgroup's time is a latency chain (inputs, weights of a channel tile, four faces, the next tile's weights, ...):
    // while the chip has room for it (up to 512 faces on 256 compute units), two workgroups share a group's channel tiles
    static std::atomic<int> cus_of[64];        // compute units per device (0 = not asked yet)
    int dev = 0, cus = 0;
    FRA_HIP(hipGetDevice(&dev));
    if (dev < 64) cus = cus_of[dev].load(std::memory_order_relaxed);
    if (cus <= 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (dev < 64) cus_of[dev].store(cus, std::memory_order_relaxed);
    }
    const dim3 grid(groups, 2 * groups <= cus ? TAIL_C / 32 / TAIL_NW : 1), block(TAIL_NW * 64);
    static std::atomic<uint64_t> lds_set[2];
    if (dt == ACT_F16) {
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(tail_kernel<f16>), TAIL_LDS, lds_set[0])) return rc;
        hipLaunchKernelGGL(tail_kernel<f16>, grid, block, TAIL_LDS, s, (const f16*)x, (const f16*)w6, b6, s6, (const f16*)w7, b7, (f16*)out, B);
    } else if (dt == ACT_BF16) {
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(tail_kernel<bf16>), TAIL_LDS, lds_set[1])) return rc;
        hipLaunchKernelGGL(tail_kernel<bf16>, grid, block, TAIL_LDS, s, (const bf16*)x, (const bf16*)w6, b6, s6, (const bf16*)w7, b7, (bf16*)out, B);
    } else {
        FRA_REQUIRE(false, FRA_ERR_ARG, "tail: unsupported activation dtype %d", dt);
    }
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}
