// Device-side pieces shared by the fused block kernels (kernels_block.hip, kernels_rblock.hip): compile-time
// loops, hand-issued LDS reads, and the depthwise + project stage that runs on the matrix cores.
#pragma once

#include <utility>

#include "common.h"

namespace fra {

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// 16-byte LDS read into `dst` from byte address `addr` (a VGPR) + compile-time `off`; the data is
// valid only after a matching counted wait
#define LDS_READ16(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---- issue order of the depthwise + project stage -------------------------------------------------
// A slice (64 expanded channels) is four 16-channel chunks q; a chunk's depthwise conv is five
// 16x16x32 MFMAs (step p = one tap pair of dw_tap(); the tenth tap has zero weights) per 16-pixel subtile,
// two subtiles per wave: step ST = 5 q + p, 20 steps.  LDS reads in issue order:
//   preamble : bias(0), slope(0), w16(0)[0..4], taps(0), taps(1)          (taps = one read per subtile)
//   step ST  : taps(ST + PD); if q odd and p == 2 the first min(4, NT16) project fragments of the pair;
//              WAIT for taps(ST); w16(q+1)[p]; after the MFMAs of p == 0: bias(q+1);
//              at p == 4 after the PReLU: slope(q+1), then for q odd the project loop
//              { WAIT fragment n; two MFMAs; issue fragment n + 4 }.
// dw_outstanding() replays that order and returns the number of reads issued after the awaited one.
constexpr int DW_STEPS = 20, DW_PD = 2, DW_RING = DW_PD + 1;   // reads run DW_PD steps ahead (3 measured no faster)
constexpr int dw_outstanding(int nt16, int nq, int kind, int st, int n, int nsub = 2) {   // kind 0: taps(st); kind 1: fragment n in step st; nsub: tap reads per step
    int issued = 7;
    int taps_idx[DW_STEPS] = {};
    int a3_idx[64] = {};
    for (int i = 0; i < DW_PD; ++i) { issued += nsub; taps_idx[i] = issued - 1; }
    for (int ST = 0; ST < 5 * nq; ++ST) {
        const int q = ST / 5, p = ST % 5;
        if (ST + DW_PD < 5 * nq) { issued += nsub; taps_idx[ST + DW_PD] = issued - 1; }
        if ((q & 1) && p == 2)
            for (int i = 0; i < cmin(4, nt16); ++i) { issued += 1; a3_idx[i] = issued - 1; }
        if (kind == 0 && st == ST) return cmin(issued - 1 - taps_idx[ST], 15);
        if (q + 1 < nq) issued += 1;                      // w16 of the next chunk
        if (p == 0 && q + 1 < nq) issued += 1;            // bias of the next chunk
        if (p == 4) {
            if (q + 1 < nq) issued += 1;                  // slope of the next chunk
            if (q & 1)
                for (int i = 0; i < nt16; ++i) {
                    if (kind == 1 && st == ST && n == i) return cmin(issued - 1 - a3_idx[i], 15);
                    if (i + 4 < nt16) { issued += 1; a3_idx[i + 4] = issued - 1; }
                }
        }
    }
    return 0;
}

// tap pairs of the five depthwise MFMAs: (0,1) (3,4) (6,7) are horizontal neighbours, (5,2) vertical ones, the
// last is tap 8 with a zero weight: its odd half reads tap 5's slot again (inside the 3x3 window, so a finite
// value) against row 12 of the weight image, which is zero.  A ds_read_b128 is served in groups of 16 lanes that mix the two
// k-groups g = 0,1 (and 2,3), so those two read the two TAPS of a pair at one 16-byte unit: the slots of a
// group then differ by 0/1 (stride 1) or an odd number (stride 2) and keep their distinct banks.
constexpr int dw_tap(int p, int half) {
    constexpr int t[5][2] = {{0, 1}, {3, 4}, {6, 7}, {5, 2}, {8, 12}};
    return t[p][half];
}
#define LDS_READ_U16(dst, addr, off) asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* GPTR;  // global source of an LDS-DMA load
typedef __attribute__((address_space(3))) void* LPTR;        // its (wave-uniform) LDS destination
__device__ __forceinline__ int cmin_rt(int a, int b) { return a < b ? a : b; }


template <int H, int S> constexpr int block_pitch() {
    return S == 2 ? (H >= 28 ? ((H + 2) | 1) : H + 2) : (H >= 14 ? (H + 2 + 15) / 16 * 16 : H + 2);
}

// ---- depthwise + project stage of one slice on the matrix cores (issue order above) -----------------
// Depthwise conv as D[ch][px] = sum over (tap, ch') of Diag[ch][(tap, ch')] . E[(tap, ch')][px]: the A
// operand of a 16x16x32 MFMA holds the weights of two taps on the diagonal (lane (m = l & 15, g = l >> 4):
// k = 8 g + j is tap dw_tap(p, g & 1), channel 8 (g >> 1) + j, non-zero only where that channel is m), the
// B operand is one 16-byte unit of the LDS image per lane.  The 72 multiply-adds per 8 channels and pixel
// leave the vector ALU; a lane ends up with channels 4g .. 4g+3 of the chunk for its pixel, which after
// bias (the MFMA's C operand), PReLU and rounding IS its half of the project MFMA's B operand for the
// chunk pair (k = 8 g + j: j < 4 from the even chunk, j >= 4 from the odd one; the project weights are
// stored in that order when their slice is staged).
// NQ: chunks of the slice this wave handles (4, or 2 = one chunk pair when two waves share a tile)
// NSUB: 16-pixel subtiles this wave handles (2, or 1 when two waves share a tile: `tb` then points at the wave's subtile)
template <typename T, int CEXP, int NT16, int PW3, int ST1, int ROWB, int NQ, int NSUB = 2> struct DwStage {
    typedef typename vec8<T>::type V8;
    V8 er[DW_RING][NSUB];          // ring: image unit of a tap pair, per subtile
    V8 a3[4];                      // ring: project fragments
    uint32_t w16[5];               // this lane's depthwise weight of each tap pair (16 bits)
    uint32_t msk[4];               // where that weight sits inside the diagonal fragment (all zero for half the lanes)
    f32x4 bias, slope;
    f32x4 acc[NSUB];
    V8 bf[NSUB];                   // project B operand of the chunk pair, per subtile
    // LDS address of this lane's unit in subtile 0: [0] tap 0 / 1 (pairs 0, 1, 2 are 0, 1, 2 image rows = ROWB bytes
    // further), [1] tap 5 / 2 (pair 4 is one row further: tap 8, and for the odd half tap 5's slot with a zero weight);
    // + ST1 for subtile 1; + 32 q
    int tb[2];
    int w2l[2];                    // this lane's 3x3 weight: [0] row 0 or 1 of the weight image (pairs 0..2), [1] row 5 or 9 (pairs 3, 4); + row offset + 32 q
    int pb;                        // depthwise bias quad of channels 4g..; + 64 q; slopes CEXP*4 further
    int w3l;                       // this lane's project weight row 16 (g & 1) + m, unit g >> 1; + 32 n PW3 + 32 pair

    template <int ST> __device__ __forceinline__ void issue_taps() {
        constexpr int q = ST / 5, p = ST % 5;
        constexpr int off = (p < 3 ? p : p - 3) * ROWB + 32 * q;
        LDS_READ16(er[ST % DW_RING][0], tb[p < 3 ? 0 : 1], off);
        if constexpr (NSUB == 2) LDS_READ16(er[ST % DW_RING][1], tb[p < 3 ? 0 : 1], ST1 + off);
    }
    template <int Q, int P> __device__ __forceinline__ void issue_w16() {
        // pairs 0..2: rows 3 P (+1 for the odd half); pair 3: rows 5 / 9 (tap 5 / a copy of tap 2); pair 4: three rows further, tap 8 / the zero row 12
        if constexpr (P < 3) LDS_READ_U16(w16[P], w2l[0], 3 * P * (CEXP * 2 + 32) + 32 * Q);
        else LDS_READ_U16(w16[P], w2l[1], (P - 3) * 3 * (CEXP * 2 + 32) + 32 * Q);
    }
    template <int Q> __device__ __forceinline__ void issue_bias() { LDS_READ16(bias, pb, 64 * Q); }
    template <int Q> __device__ __forceinline__ void issue_slope() { LDS_READ16(slope, pb, CEXP * 4 + 64 * Q); }
    template <int PAIR, int N> __device__ __forceinline__ void issue_a3() { LDS_READ16(a3[N & 3], w3l, N * 32 * PW3 + 32 * PAIR); }
    template <int PAIR, int N = 0> __device__ __forceinline__ void issue_a3_first() {
        if constexpr (N < cmin(4, NT16)) { issue_a3<PAIR, N>(); issue_a3_first<PAIR, N + 1>(); }
    }
    __device__ __forceinline__ void preamble() {
        issue_bias<0>();
        issue_slope<0>();
        issue_w16<0, 0>(); issue_w16<0, 1>(); issue_w16<0, 2>(); issue_w16<0, 3>(); issue_w16<0, 4>();
        preamble_taps<0>();
    }
    template <int ST> __device__ __forceinline__ void preamble_taps() {
        if constexpr (ST < DW_PD) { issue_taps<ST>(); preamble_taps<ST + 1>(); }
    }
    template <int ST, int PAIR, int N = 0> __device__ __forceinline__ void project(f32x4 (&pacc)[NSUB][NT16]) {
        if constexpr (N < NT16) {
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a3[N & 3]) : "n"(dw_outstanding(NT16, NQ, 1, ST, N, NSUB)));
            pacc[0][N] = mfma16(a3[N & 3], bf[0], pacc[0][N]);
            if constexpr (NSUB == 2) pacc[1][N] = mfma16(a3[N & 3], bf[1], pacc[1][N]);
            if constexpr (N + 4 < NT16) {
                __builtin_amdgcn_sched_barrier(0);         // the ring slot is free once both MFMAs are issued
                issue_a3<PAIR, N + 4>();
            }
            __builtin_amdgcn_sched_barrier(0);
            project<ST, PAIR, N + 1>(pacc);
        }
    }
    template <int ST = 0> __device__ __forceinline__ void run(f32x4 (&pacc)[NSUB][NT16]) {
        if constexpr (ST < 5 * NQ) {
            constexpr int q = ST / 5, p = ST % 5, slot = ST % DW_RING;
            if constexpr (ST + DW_PD < 5 * NQ) issue_taps<ST + DW_PD>();
            if constexpr ((q & 1) && p == 2) issue_a3_first<(q >> 1)>();
            // everything issued before taps(ST) has landed with them: w16[p], bias (p == 0), slope (p == 4)
            constexpr int OUT = dw_outstanding(NT16, NQ, 0, ST, 0, NSUB);
            if constexpr (NSUB == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(er[slot][0]), "+v"(er[slot][1]), "+v"(w16[p]) : "n"(OUT));
            else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(er[slot][0]), "+v"(w16[p]) : "n"(OUT));
            if constexpr (p == 0) asm volatile("" : "+v"(bias));      // volatile statements keep their order: uses come after the wait
            if constexpr (p == 4) asm volatile("" : "+v"(slope));
            const uint32_t wd = w16[p] | (w16[p] << 16);
            const u32x4 du = {wd & msk[0], wd & msk[1], wd & msk[2], wd & msk[3]};
            const V8 dg = __builtin_bit_cast(V8, du);
            if constexpr (q + 1 < NQ) {
                __builtin_amdgcn_sched_barrier(0);
                issue_w16<q + 1, p>();
            }
            if constexpr (p == 0) {
                acc[0] = mfma16(dg, er[slot][0], bias);
                if constexpr (NSUB == 2) acc[1] = mfma16(dg, er[slot][1], bias);
                if constexpr (q + 1 < NQ) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_bias<q + 1>();
                }
            } else {
                acc[0] = mfma16(dg, er[slot][0], acc[0]);
                if constexpr (NSUB == 2) acc[1] = mfma16(dg, er[slot][1], acc[1]);
            }
            if constexpr (p == 4) {
#pragma unroll
                for (int s = 0; s < NSUB; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float d = acc[s][i];
                        bf[s][4 * (q & 1) + i] = (T)fmaf(neg_part(d), slope[i], d);   // PReLU: slope holds slope - 1
                    }
                if constexpr (q + 1 < NQ) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_slope<q + 1>();
                }
                if constexpr (q & 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    project<ST, (q >> 1)>(pacc);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            run<ST + 1>(pacc);
        }
    }
};

}  // namespace fra
