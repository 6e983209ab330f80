// Fused inverted-residual block for gfx950:  1x1 expand + PReLU -> 3x3 depthwise + PReLU ->
// 1x1 project (+ shortcut), one kernel, the expanded tensor never leaves the CU.
// (reference: Depth_Wise.forward, face_model.py:55-65)
//
// A workgroup owns a band of TR output rows of one image.  Per 64-channel slice of the expanded
// tensor it
//   1. runs the expand GEMM on the matrix cores for every input pixel of the band (+ halo rows),
//      with the input tile's MFMA fragments held in registers for the whole kernel,
//   2. writes PReLU(expand) as 16-bit values into an LDS image [row][x+1][64 ch] whose border
//      columns and out-of-image rows stay zero (that IS the depthwise conv's zero padding),
//   3. lets every lane build, for ITS output pixel, the 3x3 depthwise result of 8 channels from
//      nine 16-byte LDS reads, and feeds those 8 values straight to the project MFMA as its B
//      fragment -- the depthwise output never exists in memory at all.
// Project accumulators live in registers across slices; the epilogue adds the shortcut and stores
// NHWC.  All parameters are read from LDS: the per-channel ones (biases, slopes, 3x3 weights) are
// staged once, the two weight-matrix slices of the next 64 channels are fetched into registers
// while the current slice computes and written to LDS behind a barrier.
//
// Channel order inside the expanded tensor: a lane of half h of the expand MFMA ends up with
// channels 8q+4h+i (q<4, i<4) of each 32-channel tile, and the project MFMA wants, for k-step s,
// element j of half h to be one fixed channel.  Taking channel 16s + 8(j>>2) + 4h + (j&3) makes
// the two coincide, so a lane's 8 values per 16-channel group travel as ONE 16-byte unit from the
// expand epilogue through LDS and the depthwise stage into the project MFMA.  Everything indexed
// by expanded channel on the depthwise/project side (bias, slope, 3x3 weights, project weight
// columns) is stored pre-permuted on the host: natural channel 16g + 8a + 4h + i sits at
// position 16g + 8h + 4a + i.  The two slope arrays hold (PReLU slope - 1): prelu(v) = v + (slope-1) min(v,0).
#include "common.h"

namespace fra {

// chunk swizzle for an LDS tile whose rows hold CH 16-byte chunks (see kernels_gemm.hip)
template <int CH> __device__ __forceinline__ int bswz(int row) {
    if (CH >= 16) return row & 15;
    return (row / (16 / CH)) & (CH - 1);
}

// 16-byte unit swizzle of the activation image: unit u of pixel slot s is stored at u ^ ((s >> 1) & 7).
// Slots are 128 bytes, so a ds_read_b128 lane group (16 lanes, 256-byte bank window) that reads
// consecutive slots sees 8 even and 8 odd slots, each set with 8 distinct XOR keys: conflict free.
__device__ __forceinline__ int eswz(int slot) { return (slot >> 1) & 7; }

template <typename T, int CIN, int CEXP, int COUT, int H, int S, int TR, bool RESID, int NW, int MINW, bool DB>
__global__ __launch_bounds__(NW * 64, MINW)
void block_kernel(const T* __restrict__ X, const T* __restrict__ W1, const float* __restrict__ b1p,
                  const float* __restrict__ s1p, const T* __restrict__ W2p, const float* __restrict__ b2p,
                  const float* __restrict__ s2p, const T* __restrict__ W3p, const float* __restrict__ b3,
                  T* __restrict__ O) {
    typedef typename vec8<T>::type V8;
    typedef typename vec4<T>::type V4;
    constexpr int NT = NW * 64;
    constexpr int W = H, Ho = H / S, Wo = Ho;
    constexpr bool WHOLE = (TR == Ho);                     // the band is the whole image: no halo rows to compute
    constexpr int RIN = (S == 1) ? TR + 2 : 2 * TR + 1;    // LDS rows (with halo)
    constexpr int RCOMP = WHOLE ? H : RIN;                 // input rows the expand GEMM runs on
    constexpr int ROFF = WHOLE ? 1 : 0;                    // LDS row of the first computed input row
    constexpr int PIN = RCOMP * W, POUT = TR * Wo;
    constexpr int MTI = (PIN + 31) / 32, MTO = (POUT + 31) / 32;
    constexpr int MTIW = (MTI + NW - 1) / NW, MTOW = (MTO + NW - 1) / NW;
    constexpr int KS1 = CIN / 16, NT3 = COUT / 32, NCH = CEXP / 64;
    constexpr int PITCH = W + 2;                           // pixel slots per LDS row (zero border each side)
    constexpr int CH1 = CIN / 8;                           // 16-byte chunks per row of the W1 slice
    // LDS map.  DB: two activation images and two copies of each weight slice, so that the expand of
    // slice c+1 and the depthwise/project of slice c run between the same pair of barriers.
    constexpr int NB = DB ? 2 : 1;
    constexpr int ES_BYTES = RIN * PITCH * 128;
    constexpr int W1_BYTES = 64 * CIN * 2, W3_BYTES = COUT * 64 * 2;
    constexpr int OFF_W1 = NB * ES_BYTES, OFF_W3 = OFF_W1 + NB * W1_BYTES, OFF_PAR = OFF_W3 + NB * W3_BYTES;
    constexpr int OFF_B1 = OFF_PAR, OFF_S1 = OFF_B1 + CEXP * 4, OFF_B2 = OFF_S1 + CEXP * 4, OFF_S2 = OFF_B2 + CEXP * 4,
                  OFF_W2 = OFF_S2 + CEXP * 4;
    constexpr int LDS_BYTES = OFF_W2 + 9 * CEXP * 2;
    constexpr int W1_V = W1_BYTES / 16, W3_V = W3_BYTES / 16;          // 16-byte vectors per slice
    constexpr int W1_PT = (W1_V + NT - 1) / NT, W3_PT = (W3_V + NT - 1) / NT;
    static_assert(CEXP % 64 == 0 && CIN % 16 == 0 && COUT % 32 == 0 && Ho % TR == 0, "shape");
    static_assert(!RESID || (S == 1 && CIN == COUT), "shortcut needs matching shapes");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int band = blockIdx.x, img = blockIdx.y;
    const int y0 = band * TR;                              // first output row of the band
    const int row0 = WHOLE ? 0 : ((S == 1) ? y0 - 1 : 2 * y0 - 1);   // image row of the first computed input row

    // ---- weight-slice staging: global -> registers (issued early) -> swizzled LDS (behind a barrier) ----
    V8 wreg1[W1_PT], wreg3[W3_PT];
    auto fetch_w1 = [&](int c) {
#pragma unroll
        for (int i = 0; i < W1_PT; ++i) {
            const int v = tid + i * NT;
            if (v < W1_V) wreg1[i] = *reinterpret_cast<const V8*>(W1 + (size_t)c * 64 * CIN + (size_t)v * 8);
        }
    };
    auto fetch_w3 = [&](int c) {
#pragma unroll
        for (int i = 0; i < W3_PT; ++i) {
            const int v = tid + i * NT;
            if (v < W3_V) wreg3[i] = *reinterpret_cast<const V8*>(W3p + (size_t)(v >> 3) * CEXP + c * 64 + (v & 7) * 8);
        }
    };
    auto store_w1 = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W1_PT; ++i) {
            const int v = tid + i * NT;
            const int row = v / CH1, ch = v - row * CH1;
            if (v < W1_V)
                *reinterpret_cast<V8*>(lds + OFF_W1 + buf * W1_BYTES + ((row * CH1 + (ch ^ bswz<CH1>(row))) << 4)) = wreg1[i];
        }
    };
    auto store_w3 = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W3_PT; ++i) {
            const int v = tid + i * NT;
            const int row = v >> 3, ch = v & 7;
            if (v < W3_V)
                *reinterpret_cast<V8*>(lds + OFF_W3 + buf * W3_BYTES + ((row * 8 + (ch ^ bswz<8>(row))) << 4)) = wreg3[i];
        }
    };
    fetch_w1(0);
    fetch_w3(0);

    // zero the activation image(s) once: borders and out-of-image rows are never written afterwards
    for (int i = tid; i < NB * ES_BYTES / 16; i += NT) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // per-channel parameters of all slices
    for (int i = tid; i < CEXP / 4; i += NT) {
        reinterpret_cast<f32x4*>(lds + OFF_B1)[i] = reinterpret_cast<const f32x4*>(b1p)[i];
        reinterpret_cast<f32x4*>(lds + OFF_S1)[i] = reinterpret_cast<const f32x4*>(s1p)[i];
        reinterpret_cast<f32x4*>(lds + OFF_B2)[i] = reinterpret_cast<const f32x4*>(b2p)[i];
        reinterpret_cast<f32x4*>(lds + OFF_S2)[i] = reinterpret_cast<const f32x4*>(s2p)[i];
    }
    for (int i = tid; i < 9 * CEXP / 8; i += NT)
        reinterpret_cast<V8*>(lds + OFF_W2)[i] = reinterpret_cast<const V8*>(W2p)[i];

    // ---- input fragments (B operand of the expand GEMM), resident for the whole kernel -------
    const T* Ximg = X + (size_t)img * H * W * CIN;
    V8 xf[MTIW][KS1];
    int in_slot[MTIW];    // LDS pixel slot of this lane's input pixel, -1 = nothing to write
#pragma unroll
    for (int t = 0; t < MTIW; ++t) {
        const int p = 32 * (wave + NW * t) + r;
        const int iy = p / W, ix = p - iy * W;
        const int gy = row0 + iy;
        const bool ok = (p < PIN) && (gy >= 0) && (gy < H);
        in_slot[t] = ok ? (iy + ROFF) * PITCH + ix + 1 : -1;
        const T* src = Ximg + ((size_t)(ok ? gy : 0) * W + (ok ? ix : 0)) * CIN + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) xf[t][ks] = *reinterpret_cast<const V8*>(src + 16 * ks);
    }

    // ---- output pixel of this lane per owned output tile ------------------------------------------
    int tap_off[MTOW][9]; // LDS byte offset of unit 0 of each of the 9 taps of this lane's output pixel
    int out_pix[MTOW];    // pixel index inside the image's output map, -1 = padding lane
#pragma unroll
    for (int t = 0; t < MTOW; ++t) {
        const int po = 32 * (wave + NW * t) + r;
        const bool ok = (wave + NW * t < MTO) && (po < POUT);
        const int pc = ok ? po : 0;
        const int oy = pc / Wo, ox = pc - oy * Wo;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int slot = (oy * S + ky) * PITCH + ox * S + kx;
                tap_off[t][ky * 3 + kx] = slot * 128 + (eswz(slot) << 4);   // unit u lives at this ^ (u << 4)
            }
        out_pix[t] = ok ? (y0 + oy) * Wo + ox : -1;
    }

    // project accumulators start at the project bias: row acc_row(g,h) of tile n3 <-> channel 32 n3 + 8q + 4h + i
    f32x16 pacc[MTOW][NT3];
#pragma unroll
    for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(b3 + 32 * n3 + 8 * q + 4 * h);
#pragma unroll
            for (int t = 0; t < MTOW; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) pacc[t][n3][4 * q + i] = b[i];
        }

    // ================= expand of slice c: E[64 ch][pixels] = PReLU(W1 . X + b1) -> LDS image eb =================
    auto expand = [&](int c, int eb, int wb) {
        char* const es = lds + eb * ES_BYTES;
        const char* const w1s = lds + OFF_W1 + wb * W1_BYTES;
#pragma unroll 1
        for (int nt = 0; nt < 2; ++nt) {
            V8 w1[KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int row = nt * 32 + r, ch = 2 * ks + h;
                w1[ks] = *reinterpret_cast<const V8*>(w1s + ((row * CH1 + (ch ^ bswz<CH1>(row))) << 4));
            }
            // bias / slope of this lane's 16 channels: groups g16 = 4c + 2nt + (q>>1), position 16 g16 + 8h + 4(q&1) + i
            f32x4 bq[4];
            const int pos1 = 16 * (4 * c + 2 * nt) + 8 * h;   // q adds 16 (q>>1) + 4 (q&1)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bq[q] = *reinterpret_cast<const f32x4*>(lds + OFF_B1 + (pos1 + 16 * (q >> 1) + 4 * (q & 1)) * 4);
#pragma unroll
            for (int t = 0; t < MTIW; ++t) {
                if (wave + NW * t >= MTI) continue;       // wave-uniform
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[4 * q + i] = bq[q][i];
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) acc = mfma32(w1[ks], xf[t][ks], acc);
                if (in_slot[t] >= 0) {
                    char* dst = es + in_slot[t] * 128;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        V8 o;
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const int q = 2 * s2 + a;
                            const f32x4 sl = *reinterpret_cast<const f32x4*>(lds + OFF_S1 + (pos1 + 16 * s2 + 4 * a) * 4);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float v = acc[4 * q + i];
                                o[4 * a + i] = (T)fmaf(fminf(v, 0.0f), sl[i], v);   // PReLU: sl = slope - 1
                            }
                        }
                        const int u = (2 * nt + s2) * 2 + h;          // 16-byte unit inside the 128-byte slot
                        *reinterpret_cast<V8*>(dst + ((u ^ eswz(in_slot[t])) << 4)) = o;
                    }
                }
            }
        }
    };

    // ================= depthwise 3x3 of slice c from LDS image eb -> project MFMA =================================
    auto dwproject = [&](int c, int eb, int wb) {
        const char* const es = lds + eb * ES_BYTES;
        const char* const w3s = lds + OFF_W3 + wb * W3_BYTES;
#pragma unroll 2
        for (int s4 = 0; s4 < 4; ++s4) {                  // 16-channel k-step of the slice
            const int pos = c * 64 + 16 * s4 + 8 * h;     // permuted position of this half's 8 channels
            const int u = 2 * s4 + h;
            const char* w2base = lds + OFF_W2 + pos * 2;
            const f32x4 b2a = *reinterpret_cast<const f32x4*>(lds + OFF_B2 + pos * 4),
                        b2b = *reinterpret_cast<const f32x4*>(lds + OFF_B2 + pos * 4 + 16);
            const f32x4 s2a = *reinterpret_cast<const f32x4*>(lds + OFF_S2 + pos * 4),
                        s2b = *reinterpret_cast<const f32x4*>(lds + OFF_S2 + pos * 4 + 16);
            V8 a3[NT3];
#pragma unroll
            for (int n3 = 0; n3 < NT3; ++n3) {
                const int row = 32 * n3 + r, ch = 2 * s4 + h;
                a3[n3] = *reinterpret_cast<const V8*>(w3s + ((row * 8 + (ch ^ bswz<8>(row))) << 4));
            }
#pragma unroll
            for (int t = 0; t < MTOW; ++t) {
                if (wave + NW * t >= MTO) continue;       // wave-uniform
                float d[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { d[j] = b2a[j]; d[4 + j] = b2b[j]; }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const V8 e = *reinterpret_cast<const V8*>(es + (tap_off[t][ky * 3 + kx] ^ (u << 4)));
                        const V8 wk = *reinterpret_cast<const V8*>(w2base + (ky * 3 + kx) * CEXP * 2);
#pragma unroll
                        for (int j = 0; j < 8; ++j) d[j] = fmaf((float)e[j], (float)wk[j], d[j]);
                    }
                V8 bf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf[j] = (T)fmaf(fminf(d[j], 0.0f), s2a[j], d[j]);               // PReLU: s2 = slope - 1
                    bf[4 + j] = (T)fmaf(fminf(d[4 + j], 0.0f), s2b[j], d[4 + j]);
                }
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3) pacc[t][n3] = mfma32(a3[n3], bf, pacc[t][n3]);
            }
        }
    };

    if (!DB) {
        // one image, one copy of the weight slices: expand | barrier | depthwise+project | barrier | restage | barrier
        store_w1(0);
        store_w3(0);
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            expand(c, 0, 0);
            if (c + 1 < NCH) { fetch_w1(c + 1); fetch_w3(c + 1); }   // in flight during the depthwise / project phase
            __syncthreads();
            dwproject(c, 0, 0);
            __syncthreads();   // every wave is done with this slice's LDS image and weights
            if (c + 1 < NCH) {
                store_w1(0);
                store_w3(0);
                __syncthreads();
            }
        }
    } else {
        // software pipeline over slices: between two barriers a wave expands slice c+1 into one image and runs
        // depthwise+project of slice c from the other.  W1 of slice c sits in buffer c&1, W3 likewise.
        store_w1(0);
        store_w3(0);
        if (NCH > 1) fetch_w1(1);
        __syncthreads();
        expand(0, 0, 0);
        if (NCH > 1) store_w1(1);
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            if (c + 2 < NCH) fetch_w1(c + 2);
            if (c + 1 < NCH) fetch_w3(c + 1);
            if (c + 1 < NCH) expand(c + 1, (c + 1) & 1, (c + 1) & 1);
            dwproject(c, c & 1, c & 1);
            // W1 buffer c&1 held slice c (last read before the previous barrier); W3 buffer (c+1)&1 held slice c-1
            if (c + 2 < NCH) store_w1(c & 1);
            if (c + 1 < NCH) store_w3((c + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue: (+ shortcut) -> NHWC store, 4 channels = 8 bytes per lane per register quad -------
    T* Oimg = O + (size_t)img * Ho * Wo * COUT;
#pragma unroll
    for (int t = 0; t < MTOW; ++t) {
        if (out_pix[t] < 0) continue;
#pragma unroll
        for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 32 * n3 + 8 * q + 4 * h;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = pacc[t][n3][4 * q + i];
                if (RESID) {
                    const V4 rr = *reinterpret_cast<const V4*>(Ximg + (size_t)out_pix[t] * CIN + n);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)rr[i];
                }
                V4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (T)v[i];
                *reinterpret_cast<V4*>(Oimg + (size_t)out_pix[t] * COUT + n) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------
template <int CIN, int CEXP, int COUT, int H, int S, int TR, bool DB>
constexpr int block_lds_bytes() {
    constexpr int RIN = (S == 1) ? TR + 2 : 2 * TR + 1;
    return (DB ? 2 : 1) * (RIN * (H + 2) * 128 + 64 * CIN * 2 + COUT * 64 * 2) + 4 * CEXP * 4 + 9 * CEXP * 2;
}

template <typename T, int CIN, int CEXP, int COUT, int H, int S, int TR, bool RESID, int NW, int MINW, bool DB>
static int launch_block_cfg(const BlockParams& p, const void* x, void* out, int B, hipStream_t s) {
    FRA_REQUIRE(p.cout == COUT, FRA_ERR_ARG, "block: cout %d, fused kernel has %d", p.cout, COUT);
    const dim3 grid((H / S) / TR, B), block(NW * 64);
    constexpr int lds = block_lds_bytes<CIN, CEXP, COUT, H, S, TR, DB>();
    auto kern = block_kernel<T, CIN, CEXP, COUT, H, S, TR, RESID, NW, MINW, DB>;
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        FRA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, s, (const T*)x, (const T*)p.w1, p.b1p, p.s1p, (const T*)p.w2p, p.b2p,
                       p.s2p, (const T*)p.w3p, p.b3, (T*)out);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

template <typename T>
static int launch_block_t(const BlockParams& p, const void* x, void* out, int B, hipStream_t s) {
    const int key = p.cin * 1000000 + p.cexp * 1000 + p.h * 10 + p.stride;
    switch (key) {
        case 64 * 1000000 + 128 * 1000 + 56 * 10 + 2:   // conv_23
            return launch_block_cfg<T, 64, 128, 64, 56, 2, 4, false, 8, 2, false>(p, x, out, B, s);
        case 64 * 1000000 + 128 * 1000 + 28 * 10 + 1:   // conv_3.*
            return launch_block_cfg<T, 64, 128, 64, 28, 1, 7, true, 8, 4, false>(p, x, out, B, s);
        case 64 * 1000000 + 256 * 1000 + 28 * 10 + 2:   // conv_34
            return launch_block_cfg<T, 64, 256, 128, 28, 2, 7, false, 8, 2, false>(p, x, out, B, s);
        case 128 * 1000000 + 256 * 1000 + 14 * 10 + 1:  // conv_4.*
            return launch_block_cfg<T, 128, 256, 128, 14, 1, 14, true, 8, 2, true>(p, x, out, B, s);
        case 128 * 1000000 + 512 * 1000 + 14 * 10 + 2:  // conv_45
            return launch_block_cfg<T, 128, 512, 128, 14, 2, 7, false, 8, 2, true>(p, x, out, B, s);
        case 128 * 1000000 + 256 * 1000 + 7 * 10 + 1:   // conv_5.*
            return launch_block_cfg<T, 128, 256, 128, 7, 1, 7, true, 8, 2, true>(p, x, out, B, s);
        default: break;
    }
    FRA_REQUIRE(false, FRA_ERR_ARG, "block: no fused kernel for cin=%d cexp=%d h=%d stride=%d", p.cin, p.cexp, p.h,
                p.stride);
    return FRA_ERR_ARG;
}

int launch_block(int dt, const BlockParams& p, const void* x, void* out, int B, hipStream_t s) {
    FRA_REQUIRE(x && out && p.w1 && p.b1p && p.s1p && p.w2p && p.b2p && p.s2p && p.w3p && p.b3, FRA_ERR_ARG,
                "block: null pointer");
    FRA_REQUIRE(B > 0 && B <= 65535, FRA_ERR_ARG, "block: batch %d not in [1,65535]", B);
    const bool want_res = p.residual != 0;
    const bool is_res_cfg = (p.stride == 1);
    FRA_REQUIRE(want_res == is_res_cfg, FRA_ERR_ARG, "block: stride %d with residual=%d has no fused kernel", p.stride,
                p.residual);
    if (dt == ACT_F16) return launch_block_t<f16>(p, x, out, B, s);
    if (dt == ACT_BF16) return launch_block_t<bf16>(p, x, out, B, s);
    FRA_REQUIRE(false, FRA_ERR_ARG, "block: unsupported activation dtype %d", dt);
    return FRA_ERR_ARG;
}

}  // namespace fra
