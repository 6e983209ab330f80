// Fused inverted-residual block as a PIPELINE inside one workgroup, without workgroup-wide barriers
// (reference: Residual / Depth_Wise.forward, face_model.py:55-75; stride 1, C -> CEXP -> C, + x).
//
// kernels_block.hip walks a band through barrier-separated stages (expand a slice | barrier | depthwise +
// project | barrier ...): every wave does the same kind of work at the same time, and a quarter of the cycles go
// to waiting at those barriers.  Here the waves of a workgroup have ROLES, and the unit that moves between them is a
// TILE of 32 consecutive pixels in row-major order (a row of 28 would leave four of a tile's 32 lanes idle; pixels are
// columns of every matrix product, so which pixels share a tile changes nothing about a value):
//   * NP producer waves: wave p expands tiles p, p + NP, ... (all CEXP channels of the tile's pixels, 32x32x16 MFMAs,
//     PReLU) into a ring of NR image rows in LDS and raises the tile's flag;
//   * NC consumer waves: wave c computes output tiles c, c + NC, ...: the 3x3 windows of a tile's pixels reach 29
//     pixels back and forth, i.e. into the two neighbouring tiles only, so it waits for the flags of tiles t - 1, t and
//     t + 1, runs the depthwise + project stage of block_stages.h with per-lane tap bases (each of a lane's two pixels
//     has its own three window rows; a row of zeros where the window leaves the image), adds bias and shortcut in
//     registers and stores its pixels; then publishes its progress, which is what lets a producer overwrite the
//     oldest ring rows.
// A wave only ever waits for the tiles it needs, by polling a word in LDS; the only barrier is the one after the
// set-up.  Both weight matrices of the block stay in LDS for the whole kernel (the ring is a few rows, not a band's
// whole expanded image), so there is no weight pipeline either.  A workgroup owns whole images and walks over
// images blockIdx.y, blockIdx.y + gridDim.y, ...; pixels are numbered through (P = 784 k + 28 y + x for the k-th
// image, ring row P / 28), and the pipeline runs on across image boundaries.
//
// Per output value the arithmetic -- MFMA shapes, operand order, roundings, (acc + x) + b3 -- is block_kernel's,
// so the two kernels give the same bits and the launcher may pick either by batch size.
#include "block_stages.h"

namespace fra {

template <int CIN, int CEXP, int COUT, int H, int NR> struct PbLds {
    static constexpr int SLOT = CEXP * 2 + 16;             // bytes per pixel slot: all expanded channels + one unit of padding
#ifndef PB_BORDER
#define PB_BORDER 1
#endif
    // pixel slots per row: a zero border slot, then the pixels; the next row's border slot is this row's right border
    // (the zero row follows the ring's last row).  Rows one slot apart in bank terms keep the 16 lanes of a tap read
    // that straddles a row end on almost distinct banks.
    static constexpr int PITCH = H + PB_BORDER;
    static constexpr int ROWB = PITCH * SLOT;
    static constexpr int OFF_ZERO = NR * ROWB;             // four slots of zeros: a window row outside the image (every lane reads the same three taps), and the right border of the ring's last row
    static constexpr int PW1 = CIN * 2 + 16, PW3 = 80, NCH = CEXP / 64;
    static constexpr int W1_BYTES = 64 * PW1, W3_BYTES = 2 * COUT * PW3, WB_BYTES = W1_BYTES + W3_BYTES;
    static constexpr int OFF_WB = (OFF_ZERO + 4 * SLOT + 15) / 16 * 16;
    static constexpr int OFF_B1 = OFF_WB + NCH * WB_BYTES, OFF_S1 = OFF_B1 + CEXP * 4, OFF_B2 = OFF_S1 + CEXP * 4,
                         OFF_S2 = OFF_B2 + CEXP * 4, OFF_W2 = OFF_S2 + CEXP * 4;
    static constexpr int NTF = 16;                         // tile flags (a ring; a word only ever grows, by NTF per use)
    static constexpr int OFF_FLAGS = OFF_W2 + dw_wimage_bytes(CEXP);   // ready[NTF] | progress[8]
    static constexpr int BYTES = OFF_FLAGS + (NTF + 8) * 4;
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

// -DPB_PRIO=n (experiments): issue priority by role.  1: consumer waves raised throughout;  2: producer waves raised
// throughout;  3: a consumer raised for its depthwise + project stage only;  4: as 3, and a producer raised for its
// expand stage (not while it polls or loads)
#ifndef PB_PRIO
#define PB_PRIO 0
#endif
#ifndef PB_SPIN_LIMIT
#define PB_SPIN_LIMIT (1 << 20)   // polls (some 0.1 s) after which a wait gives up and reports it through `fault` (common.h)
#endif

template <typename T, int CIN, int CEXP, int COUT, int H, int NP, int NC, int NR>
__global__ __launch_bounds__((NP + NC) * 64, 2)
void pblock_kernel(const T* __restrict__ X, const T* __restrict__ W1, const float* __restrict__ b1p,
                   const float* __restrict__ s1p, const T* __restrict__ W2p, const float* __restrict__ b2p,
                   const float* __restrict__ s2p, const T* __restrict__ W3p, const float* __restrict__ b3,
                   T* __restrict__ O, int B, unsigned* __restrict__ fault, int spin_limit) {
    typedef typename vec8<T>::type V8;
    typedef typename vec4<T>::type V4;
    typedef PbLds<CIN, CEXP, COUT, H, NR> L;
    constexpr int NW = NP + NC, NT = NW * 64, W = H, HW = H * W, NTF = L::NTF;
    constexpr int KS1 = CIN / 16, NT16 = COUT / 16, NCH = L::NCH, CH1 = CIN / 8;
    constexpr int SLOT = L::SLOT, ROWB = L::ROWB, PW1 = L::PW1, PW3 = L::PW3;
    constexpr int NI1 = L::W1_BYTES / 1024, NI3 = L::W3_BYTES / 1024;
    static_assert(L::W1_BYTES % 1024 == 0 && L::W3_BYTES % 1024 == 0, "weight images are whole DMA pieces");
    // the lowest tile not yet expanded (at most one past the lowest tile not yet computed, t) may always be written:
    // its last row is at most (32 t + 63) / W, and the rows NR below that are read for tiles before t only
    static_assert(NP > 0 && W * (NR - 2) > 62, "ring rows");
    static_assert(W + 1 <= 32, "a tile's windows (W + 1 pixels back and forth) stay inside the two neighbouring tiles");
    static_assert(CIN == COUT && CEXP % 64 == 0 && NC <= 8, "shape");
    static_assert(L::BYTES <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    volatile int* ready = reinterpret_cast<volatile int*>(lds + L::OFF_FLAGS);   // ready[t % NTF] = t + 1 once tile t is in the ring
    volatile int* progress = ready + NTF;                  // progress[c]: last tile consumer c has finished reading for

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
    if (tid < NTF) ready[tid] = 0;
    if (tid < 8) progress[tid] = tid < NC ? tid - NC : 0x3fffffff;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#ifdef PB_STAMPS
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PB_T(t_begin);
#endif
    const int nimg = (B - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
    const int NPX = nimg * HW;                             // pixels this workgroup handles, numbered through
    const int NTL = (NPX + 31) / 32;                       // ... in tiles of 32 (the last one may be part empty)
    // pixel P -> element offset of its first channel in X / O (CIN == COUT)
    auto pixel_offset = [&](int P) -> size_t {
        const int k = P / HW;
        return ((size_t)((int)blockIdx.y + k * (int)gridDim.y) * HW + (size_t)(P - k * HW)) * CIN;
    };

    // ================= produce: expand tile t into the ring rows of its pixels =================
    const int r = lane & 31, h = lane >> 5;
    V8 xf[KS1], xn[KS1];                                   // input fragments of the tile to expand, and of the one after it
    auto load_tile = [&](V8 (&dst)[KS1], int t) {
        const int P = cmin_rt(32 * t + r, NPX - 1);        // lanes past the last pixel load a valid one and write nothing
        const T* ptr = X + pixel_offset(P) + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) dst[ks] = *reinterpret_cast<const V8*>(ptr + 16 * ks);
    };
    typedef ExpStage<T, KS1, CEXP, PW1> Ex;
    // xf holds tile t's fragments; tnext (if < NTL) is the tile this wave expands next: requested now, moved over at the end
    auto produce = [&](int t, int tnext) {
        if (tnext < NTL) load_tile(xn, tnext);
        PB_T(t0);
        const int P = 32 * t + r;
        const int G = P / W, x = P - G * W;                // ring row (numbered through) and column of this lane's pixel
        const int wr = P < NPX ? (G % NR) * ROWB + (x + 1) * SLOT + 16 * h : -1;
        const int glast = (32 * t + 31) / W;               // last row the tile writes
        if (glast >= NR) {
            // row glast - NR is overwritten: it was read for output rows up to glast - NR + 1, whose last pixel lies in
            // tile `last`; consumer c is past it once progress[c] >= last - NC + 1 (its tiles are NC apart)
            const int need = (W * (glast - NR + 2) - 1) / 32 - NC + 1;
            for (int spin = 0;; ++spin) {
                const int v = progress[lane & 7];
                if (__builtin_amdgcn_ballot_w64(v < need) == 0) break;
                if (spin >= spin_limit) {
                    if (fault && lane == 0) *fault = 0x50420000u | (unsigned)wave;   // 'PB'
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        asm volatile("" ::: "memory");
        PB_T(t1);
        PB_ACC(0, t0, t1);
        if (PB_PRIO == 4) __builtin_amdgcn_s_setprio(1);
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
                        *reinterpret_cast<V8*>(lds + wr + 128 * c + (2 * nt + s2) * 32) = o;
                    }
                }
            });
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the tile is in LDS before its flag goes up
        if (PB_PRIO == 4) __builtin_amdgcn_s_setprio(0);
        if (lane == 0) ready[t % NTF] = t + 1;
        PB_T(t2);
        PB_ACC(1, t1, t2);
        if (tnext < NTL) {
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) xf[ks] = xn[ks];
        }
    };

    // ================= consume: output tile t from the ring rows around its pixels =================
    const int m16 = lane & 15, g4 = lane >> 4;
    f32x4 b3q[NT16];
#pragma unroll
    for (int n = 0; n < NT16; ++n) b3q[n] = *reinterpret_cast<const f32x4*>(b3 + 16 * n + 4 * g4);
    auto consume = [&](int t, int cw) {                    // cw: this wave's progress word
        // this lane's pixel of each 16-pixel subtile: global offsets, validity, tap bases of the three window rows
        size_t goff[2];
        bool ok[2];
        int rowb[2][3];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int P0 = 32 * t + 16 * st + m16;
            ok[st] = P0 < NPX;
            const int P = ok[st] ? P0 : NPX - 1;           // lanes past the last pixel compute it again and store nothing
            goff[st] = pixel_offset(P) + 4 * g4;
            const int G = P / W, x = P - G * W, y = G % H;
            const int gpart = (g4 < 2 ? g4 : 2) * SLOT;    // tap (., min(g, 2)) ...
            const int lpart = x * SLOT + gpart;            // ... of this pixel inside a ring row
            const int slot0 = (G + NR - 1) % NR;           // ring slot of the row above
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int yy = y - 1 + p;
                int slot = slot0 + p;
                slot = slot >= NR ? slot - NR : slot;
                rowb[st][p] = yy >= 0 && yy < H ? slot * ROWB + lpart : L::OFF_ZERO + gpart;
            }
        }
        // shortcut values: channels 16 n + 4 g .. of this lane's pixels
        V4 rr[2][NT16];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int n = 0; n < NT16; ++n) rr[st][n] = *reinterpret_cast<const V4*>(X + goff[st] + 16 * n);
        // wait for the tiles the windows reach into
        PB_T(t0);
        {
            const int tt = t + (lane % 3) - 1;
            const bool inside = tt >= 0 && tt < NTL;
            const int idx = (tt + NTF) % NTF, want = tt + 1;
            for (int spin = 0;; ++spin) {
                const int v = ready[idx];
                if (__builtin_amdgcn_ballot_w64(inside && v < want) == 0) break;
                if (spin >= spin_limit) {
                    if (fault && lane == 0) *fault = 0x50420000u | (unsigned)wave;   // 'PB'
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        asm volatile("" ::: "memory");
        PB_T(t1);
        PB_ACC(3, t0, t1);
        f32x4 pacc[2][NT16];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int n = 0; n < NT16; ++n) pacc[st][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (PB_PRIO >= 3) __builtin_amdgcn_s_setprio(1);
        static_for<NCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            DwStage<T, CEXP, NT16, PW3, 0, 0, 4, 2, true> dw;
            dw.set_lane(m16, g4);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                dw.tbr[p] = rowb[0][p] + 128 * c;
                dw.tbr1[p] = rowb[1][p] + 128 * c;
            }
            dw.w2l = L::OFF_W2 + (g4 & 1) * dw_wplane(CEXP) + (64 * c + m16) * 4;
            dw.pb = L::OFF_B2 + 256 * c + 16 * g4;
            dw.w3l = L::OFF_WB + c * L::WB_BYTES + L::W1_BYTES + (16 * (g4 & 1) + m16) * PW3 + 16 * (g4 >> 1);
            dw.preamble();
            dw.run(pacc);
        });
        asm volatile("" ::: "memory");
        if (PB_PRIO >= 3) __builtin_amdgcn_s_setprio(0);
        if (lane == 0) progress[cw] = t;                   // every read of the windows has returned
        PB_T(t2);
        PB_ACC(4, t1, t2);
        // epilogue in registers: (acc + x) + b3, channel for channel what the lane holds
#pragma unroll
        for (int st = 0; st < 2; ++st)
            if (ok[st]) {
#pragma unroll
                for (int n = 0; n < NT16; ++n) {
                    V4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (T)((pacc[st][n][i] + (float)rr[st][n][i]) + b3q[n][i]);
                    *reinterpret_cast<V4*>(O + goff[st] + 16 * n) = o;
                }
            }
        PB_T(t3);
        PB_ACC(5, t2, t3);
    };

    // fixed roles: waves 0 .. NP-1 produce tiles p, p + NP, ..., the others consume tiles c, c + NC, ...
    if (PB_PRIO == 1 && wave >= NP) __builtin_amdgcn_s_setprio(1);
    if (PB_PRIO == 2 && wave < NP) __builtin_amdgcn_s_setprio(1);
    if (wave < NP) {
        if (wave < NTL) load_tile(xf, wave);
        for (int t = wave; t < NTL; t += NP) produce(t, t + NP);
    } else {
        for (int t = wave - NP; t < NTL; t += NC) consume(t, wave - NP);
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
#define PB_NR 11
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
    constexpr int NP = PB_NP, NC = 8 - NP, NR = PB_NR;
    constexpr int lds = PbLds<CIN, CEXP, COUT, H, NR>::BYTES;
    auto kern = pblock_kernel<T, CIN, CEXP, COUT, H, NP, NC, NR>;
    static std::atomic<uint64_t> lds_set;      // per instantiation: devices whose LDS limit is raised
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set)) return rc;
    // one workgroup per compute unit (LDS), each walking over its images in equal shares
    const int cus = pb_compute_units();
    const int per = (B + cus - 1) / cus;
    const int groups = (B + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3(1, groups), dim3((NP + NC) * 64), lds, s, (const T*)x, (const T*)p.w1, p.b1p, p.s1p,
                       (const T*)p.w2p, p.b2p, p.s2p, (const T*)p.w3p, p.b3, (T*)out, B, fault_word(), spin_limit(PB_SPIN_LIMIT));
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

// A pipeline workgroup owns whole images, k = ceil(B / CUs) of them one after the other: about 9 + 18.6 k microseconds
// per launch, measured on 256 compute units.  The band kernels go over the chip in rounds of band-images: about
// 9.25 + 11.06 ceil(2 B / CUs) microseconds with 14-row bands, 9.25 + 5.82 ceil(4 B / CUs) with 7-row bands, whichever
// is less.  The pipeline wins where a batch leaves the bands' last round part empty and above some 640 images (193-256,
// 385-512, 641 images and more on 256 compute units).
bool pblock_pays(const BlockParams& p, int B) {
    if (!pblock_has(p)) return false;
    const int64_t cus = pb_compute_units();
    const int64_t k = (B + cus - 1) / cus, r14 = (2 * (int64_t)B + cus - 1) / cus, r7 = (4 * (int64_t)B + cus - 1) / cus;
    const int64_t bands = 925 + (1106 * r14 < 582 * r7 ? 1106 * r14 : 582 * r7);   // hundredths of a microsecond
    return 900 + 1860 * k < bands;
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
