// Residual blocks (stride 1, C -> CEXP -> C, + x) chained in registers, as FOUR WIDE WAVES per workgroup: one wave per
// SIMD with the whole 512-entry register file to itself.  (reference: Residual / Depth_Wise.forward, face_model.py:55-75)
//
// kernels_rblock.hip gives a wave one pixel tile (two image rows) and runs eight waves of 256 registers, two to a
// SIMD: the pair shares the SIMD's matrix pipe by age, every weight fragment read from LDS feeds two matrix
// instructions, every 16-pixel subtile reads all three window rows of its taps, and seven tiles leave the eighth
// wave without one.  Here a wave owns FOUR image rows of a 14 x 14 map (the four waves start at rows 0, 3, 7 and 10: rows
// 3 and 10 are computed twice, bit for bit the same, which costs what the two idle lanes of a 14-pixel row cost anyway),
// kept in registers from block to block like there, which changes the balance inside a compute unit:
//   * a weight fragment read from LDS feeds four matrix instructions instead of two;
//   * the depthwise stage walks down the wave's rows: the operand set of an image row is read once and serves the
//     three window rows of three output rows (6 row reads per 4 output rows instead of 12);
//   * four independent accumulator chains per step cover the matrix pipe's latency inside ONE instruction stream,
//     so nothing waits for a partner wave, and the PReLU of a finished tile / output row is issued between the matrix
//     instructions of the next one;
//   * four equal waves: no tile-less wave, no older / younger split;
//   * the project sums (128 registers) live in the accumulation half of the register file -- the project MFMAs are
//     inline assembly with "a" operands, every other MFMA keeps its result in ordinary registers (the file is built
//     with -mllvm -amdgpu-mfma-vgpr-form) where the PReLU that follows reads it without a copy.
// Per output value the arithmetic is kernels_rblock.hip's -- the same MFMA shapes, operand orders and roundings (expand
// on 16x16x32 with the k order the project stage's output defines, sparse 16x16x64 depthwise steps in window-row order
// on top of the bias, project sums in slice / pair order, x <- round((acc + x) + b3)) -- so the two kernels give the same
// bits, and the weight packages (RbPkg, pack_rblock) are the same bytes.
//
// Synchronisation: two workgroup barriers per slice (after the expand stage, at the end of the slice) -- four waves
// doing the same work arrive together.  -DWB_FLAGS builds the other form: words in LDS that only ever grow (slices
// done), polled at the two stage boundaries of a slice, where no hand-issued LDS read is in flight; a poll that runs
// out stores the kernel's tag to `fault` (host-visible) before going on.  Measured at batch 1024: 344 us per launch of
// six blocks with barriers, 354 with flags.  Weight package s + 1 is copied by LDS-DMA (every wave a quarter of its
// pieces) into the buffer slice s - 1 used, while slice s is computed.
//
// MEASURED AGAINST kernels_rblock.hip (same box, batch 1024, six blocks): 344 us against 306 us -- this kernel LOSES and
// is not what the forward runs (FRA_WBLOCK=1 selects it).  Its cycle stamps (-DWB_STAMPS, scripts/diag/wb_stamps.py) per
// slice and wave: expand stage 2.2k cycles for 64 MFMAs (1.0k cycles of matrix pipe), depthwise + project 3.7k for 48 + 64
// (1.8k), block epilogue 1.2k, package request 0.5k.  One wave per SIMD issues its matrix, vector and LDS instructions
// through one in-order stream: an MFMA holds the issue port for 8 of its 16 cycles, and with two PReLU instructions per
// result register, the LDS reads and the waits a depthwise step of 3 sparse MFMAs (48 pipe cycles) takes 100 to 120 issue
// cycles.  Two waves per SIMD overlap one wave's vector work with the other's matrix work; one wide wave cannot.
#include <stdlib.h>

#include "block_stages.h"

namespace fra {

#ifndef WB_SPIN_LIMIT
#define WB_SPIN_LIMIT (1 << 20)
#endif
#ifndef WB_EXP_PD
#define WB_EXP_PD 3            // expand stage: weight fragment reads run this many steps ahead of their MFMAs
#endif

// -DWB_SAFE_WAITS: every counted wait becomes a full one (a build to compare the counted build's bits with)
#ifdef WB_SAFE_WAITS
#define WB_CNT(n) 0
#else
#define WB_CNT(n) (n)
#endif

#define LDS_WRITE16(addr, data, off) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(data), "n"(off))
#define LDS_WRITE_U32(addr, data) asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(data))
// Operands of an asm statement inside a generic lambda do not make the lambda capture them: name them once in an
// ordinary expression first.
#define WB_USE(x) (void)&(x)

// ---- issue order of the expand stage (one slice): reads in program order, waits counted from it ----------------
//   preamble : bias quad of tile 0, fragments 0 .. PD - 1
//   step S = (tile tl, k-step ks): at ks == 0 the bias quad of tile tl + 1 and the slope quad of tile tl, then
//              fragment S + PD; WAIT for fragment S; the NSB MFMAs of the step, each followed by a piece of the
//              previous tile's PReLU
struct WbExSched { int w[32]; };
constexpr WbExSched wb_ex_sched(int KS, int PD) {
    WbExSched r{};
    const int NS = 4 * KS;
    int pos[32] = {};
    int issued = 1;                                        // bias quad of tile 0
    for (int i = 0; i < PD; ++i) { issued += 1; pos[i] = issued - 1; }
    for (int S = 0; S < NS; ++S) {
        const int tl = S / KS, ks = S % KS;
        if (ks == 0) issued += (tl < 3 ? 2 : 1);
        if (S + PD < NS) { issued += 1; pos[S + PD] = issued - 1; }
        r.w[S] = cmin(issued - 1 - pos[S], 15);
    }
    return r;
}

// ---- issue order of the depthwise + project stage (one slice) ------------------------------------------------------
// A slice is four 16-channel chunks q; a chunk walks over the NSB + 2 image rows R its output rows' windows touch: row
// step G = (NSB + 2) q + R.  taps(G) = the two 16-byte units of this lane's tap of row R, chunk q.
//   preamble : w(0)[0..2], slope(0), bias of output rows 0 and 1 (straight into their accumulators), taps(0), taps(1)
//   step G   : what step G + 2 needs, BEFORE its taps -- bias of output row R + 2 (R == NSB: of the next chunk's row 0,
//              R == NSB + 1: of its row 1); at R == NSB - 2: w(q + 1)[0..2]; at R == 0, q > 0: slope(q); at R == NSB - 1, q odd:
//              the first four project fragments of the chunk pair -- then taps(G + 2);
//              WAIT for taps(G); the sparse MFMAs of window rows 2, 1, 0 (output rows R - 2, R - 1, R); PReLU of output
//              row R - 2; after the last step of an odd chunk the project loop { WAIT fragment n; NSB MFMAs; issue n + 4 }.
template <int NSB> struct WbDwSched { int wt[4 * (NSB + 2)]; int wa[2][16]; };
template <int NSB, int NT16> constexpr WbDwSched<NSB> wb_dw_sched() {
    constexpr int NR = NSB + 2, NG = 4 * NR;
    WbDwSched<NSB> r{};
    int pos_t[NG] = {};
    int pos_a[16] = {};
    int issued = 3 + 1 + 2;
    issued += 2; pos_t[0] = issued - 1;
    issued += 2; pos_t[1] = issued - 1;
    for (int G = 0; G < NG; ++G) {
        const int q = G / NR, R = G % NR;
        if (R + 2 < NSB) issued += 1;
        if (R >= NSB && q < 3) issued += 1;
        if (R == NSB - 2 && q < 3) issued += 3;
        if (R == 0 && q > 0) issued += 1;
        if (R == NSB - 1 && (q & 1))
            for (int n = 0; n < cmin(4, NT16); ++n) { issued += 1; pos_a[n] = issued - 1; }
        if (G + 2 < NG) { issued += 2; pos_t[G + 2] = issued - 1; }
        r.wt[G] = cmin(issued - 1 - pos_t[G], 15);
        if (R == NR - 1 && (q & 1))
            for (int n = 0; n < NT16; ++n) {
                r.wa[q >> 1][n] = cmin(issued - 1 - pos_a[n], 15);
                if (n + 4 < NT16) { issued += 1; pos_a[n + 4] = issued - 1; }
            }
    }
    return r;
}

template <int C, int H> constexpr int wblock_lds_bytes() {
    return (H + 2) * block_pitch<H, 1>() * 144 + 2 * RbPkg<C>::BYTES + 256;
}

// project MFMA with the sums in accumulation registers
__device__ __forceinline__ void mfma16_a(f32x4& acc, vec8<f16>::type a, vec8<f16>::type b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_a(f32x4& acc, vec8<bf16>::type a, vec8<bf16>::type b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

#ifdef WB_STAMPS
__device__ unsigned long long fra_wb_cycles[65536 * 4 * 8];
extern "C" int fra_dbg_read_wb(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fra_wb_cycles), bytes, 0, hipMemcpyDeviceToHost);
}
#define WB_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define WB_ACC(i, a, b) dbg[i] += (b) - (a)
#else
#define WB_T(var)
#define WB_ACC(i, a, b)
#endif

template <typename T, int C, int CEXP, int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void wblock_kernel(const T* __restrict__ X, const char* __restrict__ WP, T* __restrict__ O, int B, int nblk,
                   unsigned* __restrict__ fault, int spin_limit) {
    typedef typename vec8<T>::type V8;
    typedef typename vec4<T>::type V4;
    typedef typename vec16<T>::type V16;
    typedef RbPkg<C> PK;
    constexpr int W = H, NW = 4, NSB = 4;                  // a wave: NSB rows = NSB subtiles of 16 lanes
    static_assert(H == 14, "a row is one 16-lane subtile; rows 0-3, 3-6, 7-10, 10-13");
    constexpr int KS = C / 32, NT16 = C / 16, NCH = CEXP / 64;
    static_assert(KS == 4, "a PReLU piece per k-step");
    constexpr int PITCH = block_pitch<H, 1>(), SLOT = 144, ROWB = PITCH * SLOT;
    constexpr int IMG_BYTES = (H + 2) * ROWB;
    constexpr int OFF_PKG = IMG_BYTES, OFF_FLAGS = OFF_PKG + 2 * PK::BYTES;
    constexpr int LDS_BYTES = OFF_FLAGS + 256;
    static_assert(LDS_BYTES == wblock_lds_bytes<C, H>() && LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((NSB + 1) * ROWB + 128 < 65536 && (NT16 - 1) * 32 * PK::PW3 + 64 < 65536, "offsets fit the instructions' field");
    constexpr int NR = NSB + 2, NG = 4 * NR;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, g4 = lane >> 4;
    const int row0 = wave == 0 ? 0 : (wave == 1 ? 3 : (wave == 2 ? 7 : 10));   // the wave's first row
    const int xx = m16 < W ? m16 : W - 1;                  // lanes 14, 15 duplicate pixel 13: same values to the same places
    const bool valid = m16 < W;
    const int npk = nblk * NCH;

    // zeroed images (border slots and border rows are never written afterwards) and flags
    for (int i = tid; i < OFF_PKG / 16; i += NW * 64) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 64) reinterpret_cast<int*>(lds + OFF_FLAGS)[tid] = 0;
    __syncthreads();

    // ---- flags: word 4 k + w = what wave w has finished, k = 0: packages landed (its pieces), 1: slices expanded,
    // 2: slices whose taps are read, 3: slices done.  Lane l polls word l & 15 against s + its delta.
    const int fl_word = OFF_FLAGS + 4 * (lane & 15);
    const int kind = (lane & 15) >> 2, who = lane & 3;
    const bool is_nb = who == wave - 1 || who == wave + 1; // the waves whose rows this wave's windows touch, and who read this wave's
#ifdef WB_FLAGS
    const int NEVER = -(1 << 30);
    const int d1 = kind == 0 ? 1 : (kind == 2 && is_nb ? 0 : NEVER);            // before expand(s)
    const int d2 = kind == 1 && is_nb ? 1 : (kind == 3 ? 0 : NEVER);            // before the depthwise stage of slice s
    auto poll = [&](int need) __attribute__((always_inline)) {
        int spin = 0;
        for (;;) {
            int v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(fl_word));
            if (__builtin_amdgcn_ballot_w64(v < need) == 0) break;
            if (++spin > spin_limit) {
                if (fault && lane == 0) *fault = 0x57420000u | (unsigned)wave;  // 'WB': a wait gave up
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
#else
    (void)fl_word; (void)kind; (void)who; (void)is_nb; (void)fault; (void)spin_limit;
#endif
    auto post = [&](int k, int value) __attribute__((always_inline)) {
        const int a = OFF_FLAGS + 4 * (4 * k + wave);
        LDS_WRITE_U32(a, value);
    };

    // ---- package pieces: piece k is copied by wave k % NW ------------------------------------------------------
    auto stage = [&](int p, int buf) __attribute__((always_inline)) {
        const char* g = WP + (size_t)p * PK::BYTES + lane * 16;
        char* l = lds + OFF_PKG + buf * PK::BYTES;
#pragma unroll
        for (int j = 0; j < (PK::PIECES + NW - 1) / NW; ++j) {
            const int k = cmin_rt(wave + NW * j, PK::PIECES - 1);   // past the end: the last piece once more, no branch
            __builtin_amdgcn_global_load_lds((GPTR)(g + k * 1024), (LPTR)(l + k * 1024), 16, 0, 0);
        }
    };

    // ---- lane constants ------------------------------------------------------------------------------------------
    // expand output: this lane's 32 bytes (positions 16 g ..) of pixel (row0 + j, xx): + j ROWB (+ 16: second unit)
    const int px0 = ((1 + row0) * PITCH + xx + 1) * SLOT + 32 * g4;
    // depthwise taps: tap (0, min(g, 2)) of output pixel (row0, xx): LDS row row0, slot xx + min(g, 2); + R ROWB + 32 q
    // (lanes 14 and 15 stay copies of lane 13 through every stage: their next block's expand stage writes pixel 13's slot)
    const int tb = (row0 * PITCH + xx + (g4 < 2 ? g4 : 2)) * SLOT;
    const int x_off = (row0 * W + xx) * C + 4 * g4;        // element offset inside the image; + j W C + 32 ks (+ 16)
    const int w1l0 = OFF_PKG + (16 * (g4 & 1) + m16) * PK::PW1 + 16 * (g4 >> 1);
    const int pl0 = OFF_PKG + PK::OFF_B1 + 16 * g4;
    const int w2l0 = OFF_PKG + PK::OFF_W2 + (g4 & 1) * dw_wplane(64) + m16 * 4;
    const int pb0 = OFF_PKG + PK::OFF_B2 + 16 * g4;
    const int w3l0 = OFF_PKG + PK::OFF_W3 + (16 * (g4 & 1) + m16) * PK::PW3 + 16 * (g4 >> 1);
    const int b3l0 = OFF_PKG + PK::OFF_B3 + 16 * g4;
    // sparse depthwise fragment: where this lane's two weights of a window row sit (DwStage::set_lane)
    uint32_t msk[4];
    int sidx;
    {
        const bool mine = (m16 >> 3) == (g4 >> 1);
        const int r = (m16 & 7) >> 2, pos = m16 & 3, other = pos ^ 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) msk[i] = !mine ? 0u : (i == r ? 0x0000FFFFu : (i == 2 + r ? 0xFFFF0000u : 0u));
        sidx = 0x4444;
        sidx = (sidx & ~(0xF << (4 * r))) | ((pos | (other << 2)) << (4 * r));
        sidx = (sidx & ~(0xF << (4 * (2 + r)))) | ((other | (pos << 2)) << (4 * (2 + r)));
    }

    V8 xf[NSB][KS];          // the activation: B operand of the expand GEMM and the shortcut
    f32x4 pacc[NSB][NT16];   // project sums: tile n, register i <-> output channel 16 n + 4 g + i

    auto load_x = [&](int gi) __attribute__((always_inline)) {
        const T* Ximg = X + (size_t)gi * H * W * C + x_off;
#pragma unroll
        for (int j = 0; j < NSB; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const V4 lo = *reinterpret_cast<const V4*>(Ximg + j * W * C + 32 * ks);
                const V4 hi = *reinterpret_cast<const V4*>(Ximg + j * W * C + 32 * ks + 16);
                xf[j][ks] = V8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
    };

    // ================= expand stage of one slice: E = PReLU(W1 . x + b1) into the wave's rows of its image =================
    auto expand = [&](int buf) __attribute__((always_inline)) {
        constexpr WbExSched SC = wb_ex_sched(KS, WB_EXP_PD);
        constexpr int PD = WB_EXP_PD, NB = PD + 1, NS = 4 * KS;
        const int w1l = w1l0 + buf * PK::BYTES, pl = pl0 + buf * PK::BYTES;
        V8 fr[NB];
        f32x4 bq[2], sl[2];
        f32x4 acc[2][NSB];
        V8 o[NSB];
        LDS_READ16(bq[0], pl, 0);
        static_for<PD>([&](auto I) {
            constexpr int S = decltype(I)::value;
            WB_USE(fr); WB_USE(w1l);
            LDS_READ16(fr[S % NB], w1l, (S / KS) * 32 * PK::PW1 + (S % KS) * 32);
        });
        // one value of the PReLU of tile t, row j: o[j][4 (t & 1) + i]
        auto piece = [&](auto t_c, auto j_c, auto i_c) {
            constexpr int t = decltype(t_c)::value, j = decltype(j_c)::value, i = decltype(i_c)::value;
            const float v = acc[t & 1][j][i];
            o[j][4 * (t & 1) + i] = (T)fmaf(neg_part(v), sl[t & 1][i], v);
        };
        auto write_pair = [&](auto p_c) {
            constexpr int p = decltype(p_c)::value;
            static_for<NSB>([&](auto J) {
                constexpr int j = decltype(J)::value;
                WB_USE(px0); WB_USE(o);
                LDS_WRITE16(px0, o[j], j * ROWB + 16 * p);
            });
        };
        static_for<NS>([&](auto SS) {
            constexpr int S = decltype(SS)::value, tl = S / KS, ks = S % KS, slot = S % NB;
            WB_USE(fr); WB_USE(w1l); WB_USE(bq); WB_USE(sl); WB_USE(pl);
            if constexpr (ks == 0) {
                if constexpr (tl < 3) LDS_READ16(bq[(tl + 1) & 1], pl, 64 * (tl + 1));
                LDS_READ16(sl[tl & 1], pl, 256 + 64 * tl);
            }
            if constexpr (S + PD < NS) LDS_READ16(fr[(S + PD) % NB], w1l, ((S + PD) / KS) * 32 * PK::PW1 + ((S + PD) % KS) * 32);
            if constexpr (ks == 0)
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fr[slot]), "+v"(bq[tl & 1]) : "n"(WB_CNT(SC.w[S])));
            else
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[slot]) : "n"(WB_CNT(SC.w[S])));
            if constexpr (tl > 0 && ks == 0) asm volatile("" : "+v"(sl[(tl - 1) & 1]));   // landed before this wait's fragment
            static_for<NSB>([&](auto J) {
                constexpr int j = decltype(J)::value;
                if constexpr (ks == 0) acc[tl & 1][j] = mfma16(fr[slot], xf[j][0], bq[tl & 1]);
                else acc[tl & 1][j] = mfma16(fr[slot], xf[j][ks], acc[tl & 1][j]);
                if constexpr (tl > 0) piece(std::integral_constant<int, tl - 1>{}, J, std::integral_constant<int, ks>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (tl == 2 && ks == KS - 1) {       // tiles 0 and 1 are complete: their unit of every row
                write_pair(std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        asm volatile("" : "+v"(sl[1]));                    // the last wait (nothing in flight) covers the slope quad of tile 3
        static_for<NSB>([&](auto J) {
            static_for<4>([&](auto I) { piece(std::integral_constant<int, 3>{}, J, I); });
        });
        write_pair(std::integral_constant<int, 1>{});
    };

    // ================= depthwise + project stage of one slice =================
    auto dwproject = [&](int buf, int s_next) __attribute__((always_inline)) {
        constexpr WbDwSched<NSB> SC = wb_dw_sched<NSB, NT16>();
        const int w2l = w2l0 + buf * PK::BYTES, pb = pb0 + buf * PK::BYTES, w3l = w3l0 + buf * PK::BYTES;
        V8 er[3][2];             // ring: the two image units of this lane's tap of a row
        V8 a3[4];                // ring: project fragments
        uint32_t wn[3];          // this lane's two depthwise weights per window row (even slot | odd slot << 16)
        V8 dg[3];                // ... as sparse fragments
        f32x4 slope;
        f32x4 acc[NSB];
        V8 bf[NSB];              // project B operand of the chunk pair, per row

        LDS_READ_U32(wn[0], w2l, 0); LDS_READ_U32(wn[1], w2l, dw_wrow(64)); LDS_READ_U32(wn[2], w2l, 2 * dw_wrow(64));
        LDS_READ16(slope, pb, 256);
        LDS_READ16(acc[0], pb, 0);
        LDS_READ16(acc[1], pb, 0);
        LDS_READ16(er[0][0], tb, 0); LDS_READ16(er[0][1], tb, 16);
        LDS_READ16(er[1][0], tb, ROWB); LDS_READ16(er[1][1], tb, ROWB + 16);

        static_for<NG>([&](auto GG) {
            constexpr int G = decltype(GG)::value, q = G / NR, R = G % NR, slot = G % 3;
            WB_USE(acc); WB_USE(pb); WB_USE(wn); WB_USE(w2l); WB_USE(slope); WB_USE(a3); WB_USE(w3l); WB_USE(er); WB_USE(tb);
            // ---- what step G + 2 needs, before its taps
            if constexpr (R + 2 < NSB) LDS_READ16(acc[R + 2], pb, 64 * q);
            if constexpr (R >= NSB && q < 3) LDS_READ16(acc[R - NSB], pb, 64 * (q + 1));
            if constexpr (R == NSB - 2 && q < 3) {
                LDS_READ_U32(wn[0], w2l, 64 * (q + 1));
                LDS_READ_U32(wn[1], w2l, dw_wrow(64) + 64 * (q + 1));
                LDS_READ_U32(wn[2], w2l, 2 * dw_wrow(64) + 64 * (q + 1));
            }
            if constexpr (R == 0 && q > 0) LDS_READ16(slope, pb, 256 + 64 * q);
            if constexpr (R == NSB - 1 && (q & 1))
                static_for<cmin(4, NT16)>([&](auto N) {
                    constexpr int n = decltype(N)::value;
                    WB_USE(a3); WB_USE(w3l);
                    LDS_READ16(a3[n], w3l, n * 32 * PK::PW3 + 32 * (q >> 1));
                });
            if constexpr (G + 2 < NG) {
                constexpr int q2 = (G + 2) / NR, R2 = (G + 2) % NR;
                LDS_READ16(er[(G + 2) % 3][0], tb, R2 * ROWB + 32 * q2);
                LDS_READ16(er[(G + 2) % 3][1], tb, R2 * ROWB + 32 * q2 + 16);
            }
            // ---- taps(G), and everything issued before them, have landed
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(er[slot][0]), "+v"(er[slot][1]) : "n"(WB_CNT(SC.wt[G])));
            if constexpr (R == 0) {
                asm volatile("" : "+v"(wn[0]), "+v"(wn[1]), "+v"(wn[2]), "+v"(acc[0]));
                static_for<3>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    const u32x4 du = {wn[p] & msk[0], wn[p] & msk[1], wn[p] & msk[2], wn[p] & msk[3]};
                    dg[p] = __builtin_bit_cast(V8, du);
                });
            }
            if constexpr (R < NSB && R > 0) asm volatile("" : "+v"(acc[R]));
            if constexpr (R == 2) asm volatile("" : "+v"(slope));
            const V16 e = __builtin_shufflevector(er[slot][0], er[slot][1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
            if constexpr (R >= 2) acc[R - 2] = smfma16(dg[2], e, acc[R - 2], sidx);
            if constexpr (R >= 1 && R <= NSB) acc[R - 1] = smfma16(dg[1], e, acc[R - 1], sidx);
            if constexpr (R < NSB) acc[R] = smfma16(dg[0], e, acc[R], sidx);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (R >= 2) {
                static_for<4>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    const float d = acc[R - 2][i];
                    bf[R - 2][4 * (q & 1) + i] = (T)fmaf(neg_part(d), slope[i], d);   // PReLU: slope holds slope - 1
                });
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (G == NG - 1) {                   // every tap read of the slice has returned
                post(2, s_next);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (R == NR - 1 && (q & 1)) {
                static_for<NT16>([&](auto N) {
                    constexpr int n = decltype(N)::value;
                    WB_USE(a3); WB_USE(w3l);
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a3[n & 3]) : "n"(WB_CNT(SC.wa[q >> 1][n])));
                    static_for<NSB>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        mfma16_a(pacc[j][n], a3[n & 3], bf[j]);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (n + 4 < NT16) {
                        LDS_READ16(a3[n & 3], w3l, (n + 4) * 32 * PK::PW3 + 32 * (q >> 1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
        });
    };

    // ---- epilogue of a block in registers: x <- round((acc + x) + b3); the last block of a launch also stores ----
    auto block_epilogue = [&](int buf, bool store, int gi) __attribute__((always_inline)) {
        const int b3l = b3l0 + buf * PK::BYTES;
        f32x4 bq3[NT16];
        static_for<NT16>([&](auto N) { constexpr int n = decltype(N)::value; WB_USE(bq3); WB_USE(b3l); LDS_READ16(bq3[n], b3l, 64 * n); });
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // ... and the last project MFMAs have written their sums
        static_for<NT16>([&](auto N) { constexpr int n = decltype(N)::value; WB_USE(bq3); asm volatile("" : "+v"(bq3[n])); });
#pragma unroll
        for (int j = 0; j < NSB; ++j)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int n = 2 * ks + (e >> 2), i = e & 3;
                    const float v = pacc[j][n][i] + (float)xf[j][ks][e];
                    xf[j][ks][e] = (T)(v + bq3[n][i]);
                    pacc[j][n][i] = 0.f;
                }
        if (store && valid) {
            T* Oimg = O + (size_t)gi * H * W * C + x_off;
#pragma unroll
            for (int j = 0; j < NSB; ++j) {
                if (j == 0 && (wave & 1)) continue;        // rows 3 and 10 are stored by the wave above
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const V8 v = xf[j][ks];
                    *reinterpret_cast<V4*>(Oimg + j * W * C + 32 * ks) = V4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<V4*>(Oimg + j * W * C + 32 * ks + 16) = V4{v[4], v[5], v[6], v[7]};
                }
            }
        }
    };

#ifdef WB_STAMPS
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    WB_T(t_begin);
#endif
    // ================= the slice loop: slices numbered through over image pairs, blocks and slices =================
    const int nimg = (B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int NS_ALL = nimg * npk;
    // packages 0 and 1 (slice s uses package s % npk in buffer s & 1)
    stage(0, 0);
    if (NS_ALL > 1) stage(1 % npk, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    post(0, 2);
#ifndef WB_FLAGS
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
#pragma unroll
    for (int j = 0; j < NSB; ++j)
#pragma unroll
        for (int n = 0; n < NT16; ++n) pacc[j][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    int s = 0;
    for (int gi = blockIdx.x; gi < B; gi += gridDim.x) {
        load_x(gi);
#pragma unroll 1
        for (int b = 0; b < nblk; ++b) {
#pragma unroll 1
            for (int c = 0; c < NCH; ++c, ++s) {
                const int buf = s & 1;
                WB_T(t0);
#ifdef WB_FLAGS
                poll(s + d1);                              // package s has landed; the neighbour has read slice s - 1's rows
#endif
                WB_T(t1);
                WB_ACC(0, t0, t1);
                expand(buf);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the rows are in LDS before the flag goes up
#ifndef WB_FLAGS
                __builtin_amdgcn_s_barrier();              // every wave's rows of slice s are in LDS, every wave is done with slice s - 1
                asm volatile("" ::: "memory");
                WB_T(t2);
                WB_ACC(1, t1, t2);
#else
                post(1, s + 1);
                WB_T(t2);
                WB_ACC(1, t1, t2);
                poll(s + d2);                              // the neighbour's rows of slice s; every wave is done with slice s - 1
#endif
                WB_T(t3);
                WB_ACC(2, t2, t3);
                const bool more = s >= 1 && s + 1 < NS_ALL;
                if (more) stage((s + 1) % npk, buf ^ 1);   // (requested from inside the project loop instead: 3 % slower)
                WB_T(t4);
                WB_ACC(3, t3, t4);
                dwproject(buf, s + 1);
                WB_T(t5);
                WB_ACC(4, t4, t5);
                if (c == NCH - 1) block_epilogue(buf, b == nblk - 1, gi);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // this wave's pieces of package s + 1; its reads of package s
#ifndef WB_FLAGS
                __builtin_amdgcn_s_barrier();              // package s + 1 has landed, nobody reads this slice's rows any more
                asm volatile("" ::: "memory");
#else
                if (more) post(0, s + 2);
                post(3, s + 1);
#endif
                WB_T(t6);
                WB_ACC(5, t5, t6);
            }
        }
    }
#ifdef WB_STAMPS
    WB_T(t_end);
    dbg[7] = t_end - t_begin;
    if (lane == 0)
        for (int i = 0; i < 8; ++i) fra_wb_cycles[((size_t)blockIdx.x * 4 + wave) * 8 + i] = dbg[i];
#endif
}

// ---------------------------------------------------------------------------------------------
template <typename T, int C, int CEXP, int H>
static int launch_wblock_cfg(const void* x, const void* wp, int nblk, void* out, int B, hipStream_t s) {
    constexpr int lds = wblock_lds_bytes<C, H>();
    void (*kern)(const T*, const char*, T*, int, int, unsigned*, int) = wblock_kernel<T, C, CEXP, H>;
    static std::atomic<uint64_t> lds_set;
    static std::atomic<int> cus_of[64];
    int dev = 0;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set, &dev)) return rc;
    int cus = dev < 64 ? cus_of[dev].load(std::memory_order_relaxed) : 0;
    if (cus <= 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (dev < 64) cus_of[dev].store(cus, std::memory_order_relaxed);
    }
    // one workgroup per compute unit, each walking over its images in equal shares
    const int units = B;
    const int per = (units + cus - 1) / cus;
    const int groups = (units + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3(groups), dim3(256), lds, s, (const T*)x, (const char*)wp, (T*)out, B, nblk,
                       fault_word(), spin_limit(WB_SPIN_LIMIT));
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

bool wblock_has(int c, int cexp, int h) { return c == 128 && cexp == 256 && h == 14; }

int launch_wblock(int dt, int c, int cexp, int h, const void* x, const void* wp, int nblk, void* out, int B, hipStream_t s) {
    FRA_REQUIRE(x && wp && out, FRA_ERR_ARG, "wblock: null pointer");
    FRA_REQUIRE(B > 0 && nblk >= 1 && nblk <= 64, FRA_ERR_ARG, "wblock: batch %d, %d blocks", B, nblk);
    FRA_REQUIRE(wblock_has(c, cexp, h), FRA_ERR_ARG, "wblock: no kernel for c=%d cexp=%d h=%d", c, cexp, h);
    FRA_REQUIRE((((uintptr_t)wp) & 15) == 0, FRA_ERR_ARG, "wblock: packages must be 16-byte aligned");
    if (dt == ACT_F16) return launch_wblock_cfg<f16, 128, 256, 14>(x, wp, nblk, out, B, s);
    if (dt == ACT_BF16) return launch_wblock_cfg<bf16, 128, 256, 14>(x, wp, nblk, out, B, s);
    FRA_REQUIRE(false, FRA_ERR_ARG, "wblock: unsupported activation dtype %d", dt);
    return FRA_ERR_ARG;
}

}  // namespace fra
