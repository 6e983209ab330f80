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
//   32 images per block; NWV waves, wave w owns channels [w E/NWV, (w+1) E/NWV); the squared row norm is
//   reduced over registers, the two lane halves and the waves (in wave order, whatever the batch).  One
//   32-channel tile per wave where the block size allows: a block streams the whole weight matrix
//   through one CU, and at batch 1 that stream (a chain of dependent loads per wave) is all there is.
//   SPLIT (a single group of up to 32 images): the same waves as NWV one-wave blocks (blockIdx.y = wave), so
//   that the weight matrix streams through NWV compute units at once; such a block cannot see the other
//   channels' squares, writes its values un-normalised, and l2norm_replay_kernel finishes the rows with the
//   same additions in the same order -- the result does not depend on which way a batch went.
// ---------------------------------------------------------------------------------------------
template <typename T, int NWV, int NTW, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 64 : NWV * 64) void linear_l2norm_kernel(const T* __restrict__ A,
                                                                            const T* __restrict__ Wt,
                                                                            const float* __restrict__ bias,
                                                                            float* __restrict__ out, int B) {
    typedef typename vec8<T>::type V8;
    constexpr int K = 512, CH = K / 8, E = 32 * NWV * NTW, NT = NWV * 64;
    __shared__ __attribute__((aligned(16))) char smem[SPLIT ? 16 : 32 * K * 2];
    __shared__ float red[SPLIT ? 1 : NWV][32];

    const int tid = threadIdx.x, lane = tid & 63, wave = SPLIT ? blockIdx.y : tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b0 = blockIdx.x * 32;
    const T* arow = A + (size_t)(b0 + r < B ? b0 + r : B - 1) * K + 8 * h;   // SPLIT: fragments straight from memory

    if constexpr (!SPLIT) {
        for (int i = tid; i < 32 * CH; i += NT) {
            const int row = i / CH, c = i - row * CH;
            int gb = b0 + row;
            gb = gb < B ? gb : B - 1;
            const V8 v = *reinterpret_cast<const V8*>(A + (size_t)gb * K + c * 8);
            *reinterpret_cast<V8*>(smem + ((row * CH + (c ^ swz<CH>(row))) << 4)) = v;
        }
        __syncthreads();
    }

    f32x16 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[nt][g] = 0.0f;
    const int nbase = wave * 32 * NTW;
    constexpr int UNR = SPLIT ? 32 : 8;   // SPLIT: every load of the wave's stream in flight at once
#pragma unroll UNR
    for (int s = 0; s < K / 16; ++s) {
        const int c = 2 * s + h;
        V8 bfrag;
        if constexpr (SPLIT) bfrag = *reinterpret_cast<const V8*>(arow + 16 * s);
        else bfrag = *reinterpret_cast<const V8*>(smem + ((r * CH + (c ^ swz<CH>(r))) << 4));
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const V8 afrag = *reinterpret_cast<const V8*>(Wt + (size_t)(nbase + nt * 32 + r) * K + s * 16 + h * 8);
            acc[nt] = mfma32(afrag, bfrag, acc[nt]);
        }
    }
    float ss = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float v = acc[nt][g] + bias[nbase + nt * 32 + acc_row(g, h)];
            acc[nt][g] = v;
            ss = fmaf(v, v, ss);
        }
    float nrm = 1.0f;
    if constexpr (!SPLIT) {
        ss += __shfl_xor(ss, 32, 64);
        if (h == 0) red[wave][r] = ss;
        __syncthreads();
        float tot = red[0][r];
#pragma unroll
        for (int w = 1; w < NWV; ++w) tot += red[w][r];
        nrm = sqrtf(tot);
    }
    const int b = b0 + r;
    if (b < B) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = SPLIT ? acc[nt][4 * q + i] : acc[nt][4 * q + i] / nrm;
                *reinterpret_cast<f32x4*>(out + (size_t)b * E + nbase + nt * 32 + 8 * q + 4 * h) = o;
            }
    }
}

// Finishes linear_l2norm_kernel<.., SPLIT>: row b of out [B][E] holds the un-normalised values.  One wave per row;
// lane 2 w + h replays what half h of wave w adds up in the un-split kernel (its tiles in order, its 16 rows of a
// tile in register order), the halves are added, the waves summed in wave order, and every value is divided by
// the root of that sum -- bit for bit what the un-split kernel writes.
template <int NWV, int NTW>
__global__ __launch_bounds__(64) void l2norm_replay_kernel(float* __restrict__ out) {
    constexpr int E = 32 * NWV * NTW;
    static_assert(2 * NWV <= 64, "one lane per (wave, half)");
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
#define FRA_LINEAR(NWV, NTW)                                                                                        \
    if (B <= 32) {                                                                                                  \
        hipLaunchKernelGGL((linear_l2norm_kernel<T, NWV, NTW, true>), dim3(1, NWV), dim3(64), 0, s, in, w, bias, out, B); \
        hipLaunchKernelGGL((l2norm_replay_kernel<NWV, NTW>), dim3(B), dim3(64), 0, s, out);                           \
    } else {                                                                                                        \
        hipLaunchKernelGGL((linear_l2norm_kernel<T, NWV, NTW, false>), grid, dim3(NWV * 64), 0, s, in, w, bias, out, B); \
    }
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
