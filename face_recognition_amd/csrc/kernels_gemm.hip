// MFMA kernels for gfx950: the pointwise (1x1) convolutions as GEMMs with a fused bias / PReLU /
// residual epilogue, and the embedding head (linear + folded BatchNorm1d + row L2 normalisation).
//
// Orientation: the weight matrix is the MFMA "A" operand (rows = output channels) and the
// activation tile the "B" operand (columns = pixels), so each lane ends up holding, for ONE
// pixel, runs of four consecutive output channels -- an 8-byte NHWC store per register quad,
// and per-channel bias / slope become per-register constants.
//
// The activation tile [BM pixels][K] is staged once per block in LDS with 16-byte chunks
// XOR-swizzled so that the fragment reads (32 rows x 16 B per half-wave) are conflict free; the
// weights (<= 128 KB, L2 resident) are read as fragments straight from global memory.
#include "common.h"

namespace fra {

// chunk swizzle for an LDS tile whose rows hold CH 16-byte chunks
template <int CH> __device__ __forceinline__ int swz(int row) {
    if (CH >= 16) return row & 15;
    return (row / (16 / CH)) & (CH - 1);
}

// ---------------------------------------------------------------------------------------------
// out[M,N] = act(in[M,K] . w[N,K]^T + bias[N]) (+ resid[M,N])      (face_model.py:51,53,62,88)
//   BM pixels per block (128, or 64 when K = 512), 4 waves laid out 2 (pixels) x 2 (channels),
//   each wave owning MT x NT accumulator tiles of 32x32; the block walks all N channels in
//   passes of 64*NT so the activation tile is read from HBM exactly once.
// ---------------------------------------------------------------------------------------------
template <typename T, int K, int BM, int NT>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const T* __restrict__ A, const T* __restrict__ Wt,
                                                      const float* __restrict__ bias,
                                                      const float* __restrict__ slope, const T* __restrict__ R,
                                                      T* __restrict__ O, long M, int N) {
    typedef typename vec8<T>::type V8;
    typedef typename vec4<T>::type V4;
    constexpr int CH = K / 8;
    constexpr int MT = BM / 64;
    __shared__ __attribute__((aligned(16))) char smem[BM * K * 2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const long m0 = (long)blockIdx.x * BM;

    for (int i = tid; i < BM * CH; i += 256) {
        const int row = i / CH, c = i - row * CH;
        long gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        const V8 v = *reinterpret_cast<const V8*>(A + gm * K + c * 8);
        *reinterpret_cast<V8*>(smem + ((row * CH + (c ^ swz<CH>(row))) << 4)) = v;
    }
    __syncthreads();

    for (int n0 = 0; n0 < N; n0 += 64 * NT) {
        f32x16 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[mt][nt][g] = 0.0f;

        const int nbase = n0 + wn * 32 * NT;
#pragma unroll
        for (int s = 0; s < K / 16; ++s) {
            V8 a[NT], b[MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                a[nt] = *reinterpret_cast<const V8*>(Wt + (size_t)(nbase + nt * 32 + r) * K + s * 16 + h * 8);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = wm * 32 * MT + mt * 32 + r;
                const int c = 2 * s + h;
                b[mt] = *reinterpret_cast<const V8*>(smem + ((row * CH + (c ^ swz<CH>(row))) << 4));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma32(a[nt], b[mt], acc[mt][nt]);
        }

#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const long m = m0 + wm * 32 * MT + mt * 32 + r;
            if (m >= M) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nbase + nt * 32 + 8 * q + 4 * h;
                    const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + n);
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[mt][nt][4 * q + i] + bi[i];
                    if (slope) {
                        const f32x4 sl = *reinterpret_cast<const f32x4*>(slope + n);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.0f ? v[i] : v[i] * sl[i];
                    }
                    if (R) {
                        const V4 rr = *reinterpret_cast<const V4*>(R + m * N + n);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += (float)rr[i];
                    }
                    V4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (T)v[i];
                    *reinterpret_cast<V4*>(O + m * N + n) = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Embedding head: out f32 [B,E] = l2norm(in[B,512] . w[E,512]^T + bias[E]).
//   (Flatten + Linear + BatchNorm1d folded + l2_norm: face_model.py:124-129, 19-22)
//   Two launches.  linear_l2norm_kernel: one wave per (group of 32 images, 32 NTW channels): its K = 512 is 32 MFMA
//   steps whose weight and activation fragments come straight from memory, all loads in flight at once, so the
//   weight matrix streams through E / (32 NTW) compute units per image group instead of one; it writes the values
//   un-normalised, because a wave sees only its own channels' squares.  l2norm_rows_kernel: one wave per image adds
//   the squares in a fixed order (per (channel group, lane half) in register order, halves, groups in order) and
//   divides.  Nothing depends on the batch size or on which images share a group.
// ---------------------------------------------------------------------------------------------
template <typename T, int NWV, int NTW>
__global__ __launch_bounds__(64) void linear_l2norm_kernel(const T* __restrict__ A, const T* __restrict__ Wt,
                                                          const float* __restrict__ bias, float* __restrict__ out, int B) {
    typedef typename vec8<T>::type V8;
    constexpr int K = 512, E = 32 * NWV * NTW;
    const int lane = threadIdx.x, cg = blockIdx.y;           // cg: channel group
    const int r = lane & 31, h = lane >> 5;
    const int b0 = blockIdx.x * 32;
    const T* arow = A + (size_t)(b0 + r < B ? b0 + r : B - 1) * K + 8 * h;

    f32x16 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[nt][g] = 0.0f;
    const int nbase = cg * 32 * NTW;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
        const V8 bfrag = *reinterpret_cast<const V8*>(arow + 16 * s);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const V8 afrag = *reinterpret_cast<const V8*>(Wt + (size_t)(nbase + nt * 32 + r) * K + s * 16 + h * 8);
            acc[nt] = mfma32(afrag, bfrag, acc[nt]);
        }
    }
    const int b = b0 + r;
    if (b < B) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = acc[nt][4 * q + i] + bias[nbase + nt * 32 + acc_row(4 * q + i, h)];
                *reinterpret_cast<f32x4*>(out + (size_t)b * E + nbase + nt * 32 + 8 * q + 4 * h) = o;
            }
    }
}

// Row b of out [B][E] holds the un-normalised values; one wave per row.  Lane 2 w + h adds the squares of channel
// group w, half h (channels 32 (w NTW + nt) + acc_row(g, h), nt then g ascending), the halves are added, the groups
// summed in order, and every value is divided by the root of that sum.
template <int NWV, int NTW>
__global__ __launch_bounds__(64) void l2norm_rows_kernel(float* __restrict__ out) {
    constexpr int E = 32 * NWV * NTW;
    static_assert(2 * NWV <= 64, "one lane per (channel group, half)");
    float* row = out + (size_t)blockIdx.x * E;
    const int lane = threadIdx.x, w = lane >> 1, h = lane & 1;
    float ss = 0.0f;
    if (w < NWV) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float v = row[(w * NTW + nt) * 32 + acc_row(g, h)];
                ss = fmaf(v, v, ss);
            }
    }
    ss += __shfl_xor(ss, 1, 64);
    float tot = __shfl(ss, 0, 64);
#pragma unroll
    for (int i = 1; i < NWV; ++i) tot += __shfl(ss, 2 * i, 64);
    const float nrm = sqrtf(tot);
    for (int j = lane; j < E; j += 64) row[j] = row[j] / nrm;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <typename T, int K, int BM>
static int launch_pw_k(const T* in, const T* w, const float* bias, const float* slope, const T* resid, T* out,
                       int64_t M, int N, hipStream_t s) {
    const int64_t blocks = (M + BM - 1) / BM;
    FRA_REQUIRE(blocks < 2147483647LL, FRA_ERR_ARG, "pw: too many rows");
    if (N % 128 == 0)
        hipLaunchKernelGGL((pw_gemm_kernel<T, K, BM, 2>), dim3((unsigned)blocks), dim3(256), 0, s, in, w, bias, slope,
                           resid, out, (long)M, N);
    else
        hipLaunchKernelGGL((pw_gemm_kernel<T, K, BM, 1>), dim3((unsigned)blocks), dim3(256), 0, s, in, w, bias, slope,
                           resid, out, (long)M, N);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

template <typename T>
static int launch_pw_t(const T* in, const T* w, const float* bias, const float* slope, const T* resid, T* out,
                       int64_t M, int K, int N, hipStream_t s) {
    switch (K) {
        case 64: return launch_pw_k<T, 64, 128>(in, w, bias, slope, resid, out, M, N, s);
        case 128: return launch_pw_k<T, 128, 128>(in, w, bias, slope, resid, out, M, N, s);
        case 256: return launch_pw_k<T, 256, 128>(in, w, bias, slope, resid, out, M, N, s);
        case 512: return launch_pw_k<T, 512, 64>(in, w, bias, slope, resid, out, M, N, s);
        default: break;
    }
    FRA_REQUIRE(false, FRA_ERR_ARG, "pw: input channels %d unsupported (64, 128, 256, 512)", K);
    return FRA_ERR_ARG;
}

int launch_pw(int dt, const void* in, const void* w, const float* bias, const float* slope, const void* resid,
              void* out, int64_t M, int K, int N, hipStream_t s) {
    FRA_REQUIRE(in && w && bias && out, FRA_ERR_ARG, "pw: null pointer");
    FRA_REQUIRE(M > 0 && N > 0 && N % 64 == 0, FRA_ERR_ARG, "pw: bad shape M=%lld N=%d (N must be a multiple of 64)",
                (long long)M, N);
    if (dt == ACT_F16)
        return launch_pw_t<f16>((const f16*)in, (const f16*)w, bias, slope, (const f16*)resid, (f16*)out, M, K, N, s);
    if (dt == ACT_BF16)
        return launch_pw_t<bf16>((const bf16*)in, (const bf16*)w, bias, slope, (const bf16*)resid, (bf16*)out, M, K,
                                 N, s);
    FRA_REQUIRE(false, FRA_ERR_ARG, "pw: unsupported activation dtype %d", dt);
    return FRA_ERR_ARG;
}

template <typename T>
static int launch_linear_t(const T* in, const T* w, const float* bias, float* out, int B, int E, hipStream_t s) {
    const dim3 grid((B + 31) / 32);
#define FRA_LINEAR(NWV, NTW)                                                                                    \
    hipLaunchKernelGGL((linear_l2norm_kernel<T, NWV, NTW>), dim3(grid.x, NWV), dim3(64), 0, s, in, w, bias, out, B); \
    hipLaunchKernelGGL((l2norm_rows_kernel<NWV, NTW>), dim3(B), dim3(64), 0, s, out);
    switch (E) {
        case 128: FRA_LINEAR(4, 1) break;
        case 256: FRA_LINEAR(8, 1) break;
        case 512: FRA_LINEAR(16, 1) break;
        case 1024: FRA_LINEAR(16, 2) break;
        default:
            FRA_REQUIRE(false, FRA_ERR_ARG, "linear: embedding_size %d unsupported (128, 256, 512, 1024)", E);
    }
#undef FRA_LINEAR
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

int launch_linear_l2norm(int dt, const void* in, const void* w, const float* bias, float* out, int B, int K, int E,
                         hipStream_t s) {
    FRA_REQUIRE(in && w && bias && out, FRA_ERR_ARG, "linear: null pointer");
    FRA_REQUIRE(B > 0 && K == 512, FRA_ERR_ARG, "linear: bad shape B=%d K=%d (K must be 512)", B, K);
    if (dt == ACT_F16) return launch_linear_t<f16>((const f16*)in, (const f16*)w, bias, out, B, E, s);
    if (dt == ACT_BF16) return launch_linear_t<bf16>((const bf16*)in, (const bf16*)w, bias, out, B, E, s);
    FRA_REQUIRE(false, FRA_ERR_ARG, "linear: unsupported activation dtype %d", dt);
    return FRA_ERR_ARG;
}

}  // namespace fra
