// Fused inverted-residual block as a ROW PIPELINE inside one workgroup, without workgroup-wide barriers
// (reference: Residual / Depth_Wise.forward, face_model.py:55-75; stride 1, C -> CEXP -> C, + x).
//
// kernels_block.hip walks a band through barrier-separated stages (expand a slice | barrier | depthwise +
// project | barrier ...): every wave does the same kind of work at the same time, and a quarter of the cycles go
// to waiting at those barriers.  Here the waves of a workgroup have ROLES:
//   * NP producer waves: wave p expands image rows p, p + NP, ... (all CEXP channels of the row, 32x32x16 MFMAs,
//     PReLU) into a ring of NR rows in LDS and raises the row's flag;
//   * NC consumer waves: wave c computes output rows c, c + NC, ...: waits for the flags of the three rows of
//     the 3x3 window, runs the depthwise + project stage of block_stages.h on them (per-row tap bases, a row of
//     zeros where the window leaves the image), adds bias and shortcut in registers and stores the row;
//     then publishes its progress, which is what lets a producer overwrite the oldest ring row.
// A wave only ever waits for the rows it needs, by polling a word in LDS; the only barrier is the one after the
// set-up.  Both weight matrices of the block stay in LDS for the whole kernel (the ring is a few rows, not a band's
// whole expanded image), so there is no weight pipeline either.  A workgroup owns whole images and walks over
// images blockIdx.y, blockIdx.y + gridDim.y, ...; rows are numbered through (G = 28 k + y for the k-th image), and
// the pipeline runs on across image boundaries.
//
// Per output value the arithmetic -- MFMA shapes, operand order, roundings, (acc + x) + b3 -- is block_kernel's,
// so the two kernels give the same bits and the launcher may pick either by batch size.
#include "block_stages.h"

namespace fra {

template <int CIN, int CEXP, int COUT, int H, int NR> struct PbLds {
    static constexpr int SLOT = CEXP * 2 + 16;             // bytes per pixel slot: all expanded channels + one unit of padding
    static constexpr int PITCH = H + 2;                    // pixel slots per row (zero border each side)
    static constexpr int ROWB = PITCH * SLOT;
    static constexpr int OFF_ZERO = NR * ROWB;             // a row of zeros (+ slack: lanes past the row end read on)
    static constexpr int PW1 = CIN * 2 + 16, PW3 = 80, NCH = CEXP / 64;
    static constexpr int W1_BYTES = 64 * PW1, W3_BYTES = 2 * COUT * PW3, WB_BYTES = W1_BYTES + W3_BYTES;
    static constexpr int OFF_WB = (OFF_ZERO + ROWB + 8 * SLOT + 15) / 16 * 16;
    static constexpr int OFF_B1 = OFF_WB + NCH * WB_BYTES, OFF_S1 = OFF_B1 + CEXP * 4, OFF_B2 = OFF_S1 + CEXP * 4,
                         OFF_S2 = OFF_B2 + CEXP * 4, OFF_W2 = OFF_S2 + CEXP * 4;
    static constexpr int OFF_FLAGS = OFF_W2 + dw_wimage_bytes(CEXP);   // ready[NR] | progress[8]
    static constexpr int BYTES = OFF_FLAGS + (NR + 8) * 4;
};

// -DPB_STAMPS: per wave, the shader-clock cycles spent in each part of its loop, summed over the kernel, go to a
// device array that fra_dbg_read() copies out (diagnostic builds only, scripts/diag/pb_stamps.py)
#ifdef PB_STAMPS
__device__ unsigned long long fra_pb_cycles[65536 * 8 * 8];
extern "C" int fra_dbg_read(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fra_pb_cycles), bytes, 0, hipMemcpyDeviceToHost);
}
#define PB_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PB_ACC(i, a, b) dbg[i] += (b) - (a)
#else
#define PB_T(var)
#define PB_ACC(i, a, b)
#endif

#ifndef PB_SPIN_LIMIT
#define PB_SPIN_LIMIT (1 << 20)   // polls (some 0.1 s) after which a wait gives up: wrong results instead of a wave that never ends
#endif

template <typename T, int CIN, int CEXP, int COUT, int H, int NP, int NC, int NR>
__global__ __launch_bounds__((NP + NC) * 64, 2)
void pblock_kernel(const T* __restrict__ X, const T* __restrict__ W1, const float* __restrict__ b1p,
                   const float* __restrict__ s1p, const T* __restrict__ W2p, const float* __restrict__ b2p,
                   const float* __restrict__ s2p, const T* __restrict__ W3p, const float* __restrict__ b3,
                   T* __restrict__ O, int B) {
    typedef typename vec8<T>::type V8;
    typedef typename vec4<T>::type V4;
    typedef PbLds<CIN, CEXP, COUT, H, NR> L;
    constexpr int NW = NP + NC, NT = NW * 64, W = H;
    constexpr int KS1 = CIN / 16, NT16 = COUT / 16, NCH = L::NCH, CH1 = CIN / 8;
    constexpr int SLOT = L::SLOT, ROWB = L::ROWB, PW1 = L::PW1, PW3 = L::PW3;
    constexpr int NI1 = L::W1_BYTES / 1024, NI3 = L::W3_BYTES / 1024;
    static_assert(L::W1_BYTES % 1024 == 0 && L::W3_BYTES % 1024 == 0, "weight images are whole DMA pieces");
    static_assert(NR >= (NP > 0 ? NC + 3 : NW + 3), "ring rows: the windows of all consumers and one row being written");
    static_assert(W <= 28 && W > 16, "a row is one 32-lane tile: two 16-pixel subtiles");
    static_assert(CIN == COUT && CEXP % 64 == 0 && NC <= 8, "shape");
    static_assert(L::BYTES <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    volatile int* ready = reinterpret_cast<volatile int*>(lds + L::OFF_FLAGS);   // ready[G % NR] = G + 1 once row G is in the ring
    volatile int* progress = ready + NR;                   // progress[c]: last output row consumer c has finished reading for

    // ---- set-up: weights of every slice by LDS-DMA, parameters, zeroed ring, flags ---------------------------
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        char* l1 = lds + L::OFF_WB + c * L::WB_BYTES;
        const T* g1 = W1 + c * 64 * CIN;
        const T* g3 = W3p + c * (2 * COUT * 32);
#pragma unroll
        for (int j = 0; j < (NI1 + NW - 1) / NW; ++j) {
            const int k = wave + NW * j;
            if (k < NI1) {                                 // wave-uniform
                const int G = 64 * k + lane, row = G / (CH1 + 1), col = cmin_rt(G - row * (CH1 + 1), CH1 - 1);
                __builtin_amdgcn_global_load_lds((GPTR)(g1 + (unsigned)(row * CIN + col * 8)), (LPTR)(l1 + k * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < (NI3 + NW - 1) / NW; ++j) {
            const int k = wave + NW * j;
            if (k < NI3) {
                const int G = 64 * k + lane, row = G / 5, u = cmin_rt(G - row * 5, 3);
                __builtin_amdgcn_global_load_lds((GPTR)(g3 + (unsigned)(row * 32 + u * 8)), (LPTR)(l1 + L::W1_BYTES + k * 1024), 16, 0, 0);
            }
        }
    }
    for (int i = tid; i < CEXP / 4; i += NT) {
        reinterpret_cast<f32x4*>(lds + L::OFF_B1)[i] = reinterpret_cast<const f32x4*>(b1p)[i];
        reinterpret_cast<f32x4*>(lds + L::OFF_S1)[i] = reinterpret_cast<const f32x4*>(s1p)[i];
        reinterpret_cast<f32x4*>(lds + L::OFF_B2)[i] = reinterpret_cast<const f32x4*>(b2p)[i];
        reinterpret_cast<f32x4*>(lds + L::OFF_S2)[i] = reinterpret_cast<const f32x4*>(s2p)[i];
    }
    for (int i = tid; i < 3 * CEXP; i += NT) {             // 3x3 weight image: window row, position -> its two words
        const int row = i / CEXP, pos = i - row * CEXP;
        const uint16_t* w2 = reinterpret_cast<const uint16_t*>(W2p) + 3 * row * CEXP + pos;
        char* dst = lds + L::OFF_W2 + row * dw_wrow(CEXP) + pos * 4;
        *reinterpret_cast<uint32_t*>(dst) = (uint32_t)w2[0] | ((uint32_t)w2[CEXP] << 16);
        *reinterpret_cast<uint32_t*>(dst + dw_wplane(CEXP)) = (uint32_t)w2[2 * CEXP];
    }
    // ring + zero row + slack: border slots stay zero for ever, everything a lane may read is a finite value
    for (int i = tid; i < L::OFF_WB / 16; i += NT) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < NR) ready[tid] = 0;
    if (tid < 8) progress[tid] = tid < (NP > 0 ? NC : NW) ? tid - (NP > 0 ? NC : NW) : 0x3fffffff;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#ifdef PB_STAMPS
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PB_T(t_begin);
#endif
    const int nimg = (B - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
    const int NG = nimg * H;                               // rows this workgroup handles, numbered through

    // ================= produce: expand row G into ring slot G % NR =================
    constexpr int NCONS = NP > 0 ? NC : NW;                // waves that consume rows (their progress words)
    const int r = lane & 31, h = lane >> 5;
    const int px = r < W ? r : W - 1;                      // lanes past the row end load a valid pixel and write nothing
    const int wr = r < W ? (r + 1) * SLOT + 16 * h : -1;
    V8 xf[KS1], xn[KS1];                                   // input fragments of the row to expand, and of the one after it
    auto load_row = [&](V8 (&dst)[KS1], int G) {
        const int k = G / H, y = G - k * H;
        const T* ptr = X + ((size_t)((int)blockIdx.y + k * (int)gridDim.y) * H * W + (size_t)y * W + px) * CIN + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) dst[ks] = *reinterpret_cast<const V8*>(ptr + 16 * ks);
    };
    typedef ExpStage<T, KS1, CEXP, PW1> Ex;
    // xf holds row G's fragments; Gnext (if < NG) is the row this wave expands next: requested now, moved over at the end
    auto produce = [&](int G, int Gnext) {
        if (Gnext < NG) load_row(xn, Gnext);
        PB_T(t0);
        if (G >= NR) {
            // slot G % NR holds row G - NR, read for output rows up to G - NR + 1: every consumer must be past them
            const int need = G - NR - NCONS + 2;
            for (int spin = 0; spin < PB_SPIN_LIMIT; ++spin) {
                const int v = progress[lane & 7];
                if (__builtin_amdgcn_ballot_w64(v < need) == 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        asm volatile("" ::: "memory");
        PB_T(t1);
        PB_ACC(0, t0, t1);
        char* rowl = lds + (G % NR) * ROWB;
        static_for<NCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            static_for<2>([&](auto ntc) {
                constexpr int nt = decltype(ntc)::value;
                Ex ex;
                ex.w1l = L::OFF_WB + c * L::WB_BYTES + r * PW1 + 16 * h;
                ex.pl = L::OFF_B1 + 256 * c + 32 * h;
                ex.template issue<nt>();
                f32x16 acc;
                ex.template chain<true>(acc, xf);
                if (wr >= 0) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        V8 o;
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const int q = 2 * s2 + a;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float v = acc[4 * q + i];
                                o[4 * a + i] = (T)fmaf(neg_part(v), ex.sl[q][i], v);   // PReLU: sl = slope - 1
                            }
                        }
                        *reinterpret_cast<V8*>(rowl + wr + 128 * c + (2 * nt + s2) * 32) = o;
                    }
                }
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the row is in LDS before its flag goes up
        if (lane == 0) ready[G % NR] = G + 1;
        PB_T(t2);
        PB_ACC(1, t1, t2);
        if (Gnext < NG) {
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) xf[ks] = xn[ks];
        }
    };

    // ================= consume: output row G from ring rows G - 1, G, G + 1 =================
    const int m16 = lane & 15, g4 = lane >> 4;
    constexpr int ST1 = 16 * SLOT;
    const int lpart = (m16 + (g4 < 2 ? g4 : 2)) * SLOT;    // this lane's tap (., min(g, 2)) of subtile 0 inside a row
    const bool ok1 = m16 + 16 < W;                         // subtile 1: columns 16 ..
    f32x4 b3q[NT16];
#pragma unroll
    for (int n = 0; n < NT16; ++n) b3q[n] = *reinterpret_cast<const f32x4*>(b3 + 16 * n + 4 * g4);
    auto consume = [&](int G, int cw) {                    // cw: this wave's progress word
        const int k = G / H, y = G - k * H;
        const size_t pix = ((size_t)((int)blockIdx.y + k * (int)gridDim.y) * H + y) * W;
        const T* xs = X + (pix + m16) * CIN + 4 * g4;
        T* os = O + (pix + m16) * COUT + 4 * g4;
        // shortcut values: channels 16 n + 4 g .. of this lane's pixel, per subtile
        V4 rr[2][NT16];
#pragma unroll
        for (int n = 0; n < NT16; ++n) {
            rr[0][n] = *reinterpret_cast<const V4*>(xs + 16 * n);
            if (ok1) rr[1][n] = *reinterpret_cast<const V4*>(xs + 16 * CIN + 16 * n);
        }
        // wait for the window's rows (those inside the image)
        PB_T(t0);
        {
            const int d = (lane % 3) - 1, yy = y + d;
            const bool inside = yy >= 0 && yy < H;
            const int idx = (G + d + NR) % NR, want = G + d + 1;
            for (int spin = 0; spin < PB_SPIN_LIMIT; ++spin) {
                const int v = ready[idx];
                if (__builtin_amdgcn_ballot_w64(inside && v < want) == 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        asm volatile("" ::: "memory");
        PB_T(t1);
        PB_ACC(3, t0, t1);
        int rowb[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int yy = y - 1 + p;
            rowb[p] = (yy >= 0 && yy < H ? ((G - 1 + p + NR) % NR) * ROWB : L::OFF_ZERO) + lpart;
        }
        f32x4 pacc[2][NT16];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int n = 0; n < NT16; ++n) pacc[st][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        static_for<NCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            DwStage<T, CEXP, NT16, PW3, ST1, 0, 4, 2> dw;
            dw.set_lane(m16, g4);
#pragma unroll
            for (int p = 0; p < 3; ++p) dw.tbr[p] = rowb[p] + 128 * c;
            dw.w2l = L::OFF_W2 + (g4 & 1) * dw_wplane(CEXP) + (64 * c + m16) * 4;
            dw.pb = L::OFF_B2 + 256 * c + 16 * g4;
            dw.w3l = L::OFF_WB + c * L::WB_BYTES + L::W1_BYTES + (16 * (g4 & 1) + m16) * PW3 + 16 * (g4 >> 1);
            dw.preamble();
            dw.run(pacc);
        });
        asm volatile("" ::: "memory");
        if (lane == 0) progress[cw] = G;                   // every read of the window has returned
        PB_T(t2);
        PB_ACC(4, t1, t2);
        // epilogue in registers: (acc + x) + b3, channel for channel what the lane holds
#pragma unroll
        for (int n = 0; n < NT16; ++n) {
            V4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (T)((pacc[0][n][i] + (float)rr[0][n][i]) + b3q[n][i]);
            *reinterpret_cast<V4*>(os + 16 * n) = o;
        }
        if (ok1) {
#pragma unroll
            for (int n = 0; n < NT16; ++n) {
                V4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (T)((pacc[1][n][i] + (float)rr[1][n][i]) + b3q[n][i]);
                *reinterpret_cast<V4*>(os + 16 * COUT + 16 * n) = o;
            }
        }
        PB_T(t3);
        PB_ACC(5, t2, t3);
    };

    if constexpr (NP > 0) {
        // fixed roles: waves 0 .. NP-1 produce rows p, p + NP, ..., the others consume rows c, c + NC, ...
        if (wave < NP) {
            if (wave < NG) load_row(xf, wave);
            for (int G = wave; G < NG; G += NP) produce(G, G + NP);
        } else {
            for (int G = wave - NP; G < NG; G += NC) consume(G, wave - NP);
        }
    } else {
        // every wave alternates: in round i it expands row NW i + w and computes output row NW i + w - 2 (the rows it
        // needs, up to NW i + w - 1, come from the same round's earlier half).  The lower half of the waves expands first
        // and computes second, the upper half the other way round (computing the previous round's row first), so the
        // two waves of a SIMD (w and w + NW / 2) do different kinds of work at any time.
        const int ph = wave >= NW / 2 ? 1 : 0;
        const int cw = (wave + NW - 2) % NW;               // rows NW i + w - 2 are the rows cw, cw + NW, ...
        if (wave < NG) load_row(xf, wave);
        const int rounds = (NG + NW - 1) / NW + 2;
        for (int t = 0; t < 2 * rounds; ++t) {
            const int i = t >> 1;
            if (((t + ph) & 1) == 0) {
                const int G = NW * i + wave;
                if (G < NG) produce(G, G + NW);
            } else {
                const int G = NW * (i - ph) + wave - 2;
                if (G >= 0 && G < NG) consume(G, cw);
            }
        }
    }
#ifdef PB_STAMPS
    PB_T(t_end);
    dbg[7] = t_end - t_begin;
    if (lane == 0)
        for (int i = 0; i < 8; ++i) fra_pb_cycles[((size_t)blockIdx.y * 8 + wave) * 8 + i] = dbg[i];
#endif
}

// ---------------------------------------------------------------------------------------------
#ifndef PB_NP
#define PB_NP 3
#endif
#ifndef PB_NR
#define PB_NR 10
#endif
static int pb_compute_units() {
    static std::atomic<int> cus_of[64];        // compute units per device (0 = not asked yet)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev < 64 && (cus = cus_of[dev].load(std::memory_order_relaxed)) > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (dev < 64) cus_of[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

template <typename T, int CIN, int CEXP, int COUT, int H>
static int launch_pblock_cfg(const BlockParams& p, const void* x, void* out, int B, hipStream_t s) {
    constexpr int NP = PB_NP, NC = 8 - NP, NR = PB_NR;    // NP == 0: every wave expands and computes in turn
    constexpr int lds = PbLds<CIN, CEXP, COUT, H, NR>::BYTES;
    auto kern = pblock_kernel<T, CIN, CEXP, COUT, H, NP, NC, NR>;
    static std::atomic<uint64_t> lds_set;      // per instantiation: devices whose LDS limit is raised
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set)) return rc;
    // one workgroup per compute unit (LDS), each walking over its images in equal shares
    const int cus = pb_compute_units();
    const int per = (B + cus - 1) / cus;
    const int groups = (B + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3(1, groups), dim3((NP + NC) * 64), lds, s, (const T*)x, (const T*)p.w1, p.b1p, p.s1p,
                       (const T*)p.w2p, p.b2p, p.s2p, (const T*)p.w3p, p.b3, (T*)out, B);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

// A pipeline workgroup owns whole images: k = ceil(B / CUs) images one after the other, about 9 + 20.6 k microseconds
// measured on 256 compute units, against 7 + 0.0906 B for the band kernels, whose bands fill the chip at any batch.  The
// pipeline wins where the batch is a little under a multiple of the compute-unit count (250-256, 477-512, 704-768,
// 931-1024 images, ...).
bool pblock_pays(const BlockParams& p, int B) {
    if (!pblock_has(p)) return false;
    const int cus = pb_compute_units();
    const int64_t k = (B + cus - 1) / cus;
    return (int64_t)B * 256 > (227 * k + 22) * cus;
}

bool pblock_has(const BlockParams& p) {
    return p.cin == 64 && p.cexp == 128 && p.cout == 64 && p.h == 28 && p.stride == 1 && p.residual != 0;
}

int launch_pblock(int dt, const BlockParams& p, const void* x, void* out, int B, hipStream_t s) {
    FRA_REQUIRE(pblock_has(p), FRA_ERR_ARG, "pblock: no pipelined kernel for cin=%d cexp=%d h=%d stride=%d", p.cin, p.cexp, p.h, p.stride);
    FRA_REQUIRE(x && out && p.w1 && p.b1p && p.s1p && p.w2p && p.b2p && p.s2p && p.w3p && p.b3, FRA_ERR_ARG, "pblock: null pointer");
    FRA_REQUIRE(B > 0 && B <= 65535, FRA_ERR_ARG, "pblock: batch %d not in [1,65535]", B);
    if (dt == ACT_F16) return launch_pblock_cfg<f16, 64, 128, 64, 28>(p, x, out, B, s);
    if (dt == ACT_BF16) return launch_pblock_cfg<bf16, 64, 128, 64, 28>(p, x, out, B, s);
    FRA_REQUIRE(false, FRA_ERR_ARG, "pblock: unsupported activation dtype %d", dt);
    return FRA_ERR_ARG;
}

}  // namespace fra
