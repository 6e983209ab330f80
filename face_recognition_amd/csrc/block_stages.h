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
// A slice (64 expanded channels) is four 16-channel chunks q; a chunk's depthwise conv is three sparse
// 16x16x64 MFMAs (step p = one row of the 3x3 window: its three taps and one empty tap slot) per 16-pixel
// subtile, one or two subtiles per wave: step ST = 3 q + p, 12 steps.  LDS reads in issue order:
//   preamble : bias(0) (one read per subtile, straight into the accumulators), slope(0), w(0)[0..2],
//              taps(0) .. taps(DW_PD - 1)                                   (taps = two reads per subtile)
//   step ST  : taps(ST + DW_PD); if q odd and p == 0 the first min(4, NT16) project fragments of the pair;
//              WAIT for taps(ST); w(q+1)[p]; after the MFMAs of p == 0: bias(q+1);
//              at p == 2 after the PReLU: slope(q+1), then for q odd the project loop
//              { WAIT fragment n; its MFMAs; issue fragment n + 4 }.
// dw_outstanding() replays that order and returns the number of reads issued after the awaited one.
#ifndef DW_PD_STEPS
#define DW_PD_STEPS 1
#endif
constexpr int DW_ROWS = 3, DW_PD = DW_PD_STEPS, DW_RING = DW_PD + 1;   // tap reads run DW_PD steps ahead
static_assert(DW_PD >= 1 && DW_PD <= 2, "w(q+1)[p] and bias(q+1) must be issued before the taps of the step that uses them");
constexpr int dw_outstanding(int nt16, int nq, int kind, int st, int n, int nsub = 2, int pd = DW_PD) {   // kind 0: taps(st); kind 1: fragment n in step st
    const int tap_reads = 2 * nsub;
    int issued = nsub + 1 + DW_ROWS;
    int taps_idx[32] = {};
    int a3_idx[64] = {};
    for (int i = 0; i < pd; ++i) { issued += tap_reads; taps_idx[i] = issued - 1; }
    for (int ST = 0; ST < DW_ROWS * nq; ++ST) {
        const int q = ST / DW_ROWS, p = ST % DW_ROWS;
        if (ST + pd < DW_ROWS * nq) { issued += tap_reads; taps_idx[ST + pd] = issued - 1; }
        if ((q & 1) && p == 0)
            for (int i = 0; i < cmin(4, nt16); ++i) { issued += 1; a3_idx[i] = issued - 1; }
        if (kind == 0 && st == ST) return cmin(issued - 1 - taps_idx[ST], 15);
        if (q + 1 < nq) issued += 1;                      // w of the next chunk
        if (p == 0 && q + 1 < nq) issued += nsub;         // bias of the next chunk
        if (p == DW_ROWS - 1) {
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

#define LDS_READ_U32(dst, addr, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* GPTR;  // global source of an LDS-DMA load
typedef __attribute__((address_space(3))) void* LPTR;        // its (wave-uniform) LDS destination
__device__ __forceinline__ int cmin_rt(int a, int b) { return a < b ? a : b; }


template <int H, int S> constexpr int block_pitch() {
    return S == 2 ? (H >= 28 ? ((H + 2) | 1) : H + 2) : (H >= 14 ? (H + 2 + 15) / 16 * 16 : H + 2);
}

// ---- expand stage: LDS reads of one 32-channel tile nt, in issue order: bias quads bq[4], fragments
// w1[KS1], slope quads sl[4]; the first pixel tile waits with counts, later tiles find the data there ----
template <typename T, int KS1, int CEXP, int PW1> struct ExpStage {
    typedef typename vec8<T>::type V8;
    V8 w1[KS1];
    f32x4 bq[4], sl[4];
    int w1l;   // LDS address of this lane's W1 row (+ 16 h); + 32 nt PW1 + 32 ks
    int pl;    // LDS address of this lane's bias quads of the slice; + 128 nt + 64 (q>>1) + 16 (q&1); slopes CEXP*4 further

    template <int NTI, int I = 0> __device__ __forceinline__ void issue() {
        if constexpr (I < 4) {
            LDS_READ16(bq[I], pl, 128 * NTI + 64 * (I >> 1) + 16 * (I & 1));
            issue<NTI, I + 1>();
        } else if constexpr (I < 4 + KS1) {
            LDS_READ16(w1[I - 4], w1l, NTI * 32 * PW1 + (I - 4) * 32);
            issue<NTI, I + 1>();
        } else if constexpr (I < 8 + KS1) {
            constexpr int q = I - 4 - KS1;
            LDS_READ16(sl[q], pl, CEXP * 4 + 128 * NTI + 64 * (q >> 1) + 16 * (q & 1));
            issue<NTI, I + 1>();
        }
    }
    // acc = bias + sum_ks w1[ks] . x[ks]
    template <bool WAIT, int KS = 0> __device__ __forceinline__ void chain(f32x16& acc, const V8 (&x)[KS1]) {
        if constexpr (KS == 0) {
            // the four bias quads and fragment 0 have arrived once KS1 - 1 + 4 reads remain
            if constexpr (WAIT)
                asm volatile("s_waitcnt lgkmcnt(%5)"
                             : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(w1[0]) : "n"(KS1 - 1 + 4));
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[4 * q + i] = bq[q][i];
            acc = mfma32(w1[0], x[0], acc);
            __builtin_amdgcn_sched_barrier(0);             // keep each MFMA right behind its wait
            chain<WAIT, 1>(acc, x);
        } else if constexpr (KS < KS1) {
            if constexpr (WAIT) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w1[KS]) : "n"(KS1 - 1 - KS + 4));
            acc = mfma32(w1[KS], x[KS], acc);
            __builtin_amdgcn_sched_barrier(0);
            chain<WAIT, KS + 1>(acc, x);
        } else {
            if constexpr (WAIT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sl[0]), "+v"(sl[1]), "+v"(sl[2]), "+v"(sl[3]));
        }
    }
};

// ---- depthwise + project stage of one slice on the matrix cores (issue order above) -----------------
// Depthwise conv as D[ch][px] = sum over (tap, ch') of Diag[ch][(tap, ch')] . E[(tap, ch')][px] on the SPARSE
// 16x16x64 MFMA (A is 2:4 structured: per row and group of four k two values and their 2-bit positions; a
// diagonal has one value per group at most).  Operand maps (lane l, m = l & 15, g = l >> 4):
//   B: element j < 8 is k = 8 g + j, element j >= 8 is k = 32 + 8 g + (j - 8);
//   A: group r < 4 covers k = 16 g + 4 r .. + 3: elements 2 r, 2 r + 1 sit at the positions bits [1:0], [3:2] of
//      nibble r of the index register select;
//   D: register i is row 4 g + i, column m.
// k-block kb = k >> 3 is DEFINED as tap slot kb & 3 of the step's window row, 8-channel unit kb >> 2 of the chunk:
// a B lane then holds the two units of ONE tap (32 consecutive bytes of a pixel slot), tap slot g; an A lane holds
// tap slots 2 (g & 1) and 2 (g & 1) + 1 of unit g >> 1, non-zero only where channel 8 (g >> 1) + (k & 7) is its
// row m: the even slot's weight as the low half of register (m & 7) >> 2, position m & 3 first; the odd slot's as
// the high half of register 2 + ((m & 7) >> 2), position m & 3 second.  Both come out of ONE 32-bit LDS word
// (even slot | odd slot << 16) by four AND masks.  Slots 0..2 are the taps (p, 0..2); slot 3 has weight zero
// and reads tap (p, 2)'s pixel again (a finite value).  The 72 multiply-adds per 8 channels and pixel
// leave the vector ALU; a lane ends up with channels 4g .. 4g+3 of the chunk for its pixel, which after
// bias (the MFMA's C operand), PReLU and rounding IS its half of the project MFMA's B operand for the
// chunk pair (k = 8 g + j: j < 4 from the even chunk, j >= 4 from the odd one; the project weights are
// stored in that order when their slice is staged).
// NQ: chunks of the slice this wave handles (4, or 2 = one chunk pair when two waves share a tile)
// NSUB: 16-pixel subtiles this wave handles (2, or 1 when two waves share a tile: `tb` then points at the wave's subtile)
// SEP: the second subtile is not a fixed distance from the first: it has tap bases of its own (tbr1, with ROWB == 0)
// PD: steps the tap reads run ahead of their MFMAs (ring of PD + 1 register sets)
template <typename T, int CEXP, int NT16, int PW3, int ST1, int ROWB, int NQ, int NSUB = 2, bool SEP = false, int PD = DW_PD> struct DwStage {
    typedef typename vec8<T>::type V8;
    typedef typename vec16<T>::type V16;
    static constexpr int DW_RING = PD + 1;
    static_assert(PD >= 1 && PD <= 2, "w(q+1)[p] and bias(q+1) must be issued before the taps of the step that uses them");
    V8 er[DW_RING][NSUB][2];       // ring: the two image units of this lane's tap, per subtile
    V8 a3[4];                      // ring: project fragments
    uint32_t w[DW_ROWS];           // this lane's two depthwise weights of each window row (even slot | odd slot << 16)
    uint32_t msk[4];               // where they sit inside the sparse fragment (all zero for half the lanes)
    int idx;                       // positions of the sparse fragment's elements
    f32x4 slope;
    f32x4 acc[2][NSUB];            // by chunk parity: the next chunk's bias lands while this chunk accumulates
    V8 bf[NSUB];                   // project B operand of the chunk pair, per subtile
    int tb;                        // LDS address of this lane's tap (0, min(g, 2)) in subtile 0; + p ROWB + 32 q (+ 16: second unit) (+ ST1: subtile 1)
    int tbr[DW_ROWS];              // (ROWB == 0: the window's rows are not a fixed distance apart) that address per window row p
    int tbr1[DW_ROWS];             // (SEP) the same for subtile 1
    int w2l;                       // this lane's word of the 3x3 weight image: plane g & 1, position m; + p dw_wrow + 64 q
    int pb;                        // depthwise bias quad of channels 4g..; + 64 q; slopes CEXP*4 further
    int w3l;                       // this lane's project weight row 16 (g & 1) + m, unit g >> 1; + 32 n PW3 + 32 pair

    // lane constants: m = lane & 15, g = lane >> 4
    __device__ __forceinline__ void set_lane(int m, int g) {
        const bool mine = (m >> 3) == (g >> 1);
        const int r = (m & 7) >> 2, pos = m & 3, other = pos ^ 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) msk[i] = !mine ? 0u : (i == r ? 0x0000FFFFu : (i == 2 + r ? 0xFFFF0000u : 0u));
        // nibble r: the value first, a zero second; nibble 2 + r: a zero first, the value second; others: two zeros
        idx = 0x4444;
        idx = (idx & ~(0xF << (4 * r))) | ((pos | (other << 2)) << (4 * r));
        idx = (idx & ~(0xF << (4 * (2 + r)))) | ((other | (pos << 2)) << (4 * (2 + r)));
    }
    template <int ST> __device__ __forceinline__ void issue_taps() {
        constexpr int q = ST / DW_ROWS, p = ST % DW_ROWS;
        constexpr int off = p * ROWB + 32 * q;
        const int base = ROWB ? tb : tbr[p];
        LDS_READ16(er[ST % DW_RING][0][0], base, off);
        LDS_READ16(er[ST % DW_RING][0][1], base, off + 16);
        if constexpr (NSUB == 2 && SEP) {
            static_assert(!SEP || ROWB == 0, "separate subtile bases come per window row");
            LDS_READ16(er[ST % DW_RING][1][0], tbr1[p], off);
            LDS_READ16(er[ST % DW_RING][1][1], tbr1[p], off + 16);
        } else if constexpr (NSUB == 2) {
            LDS_READ16(er[ST % DW_RING][1][0], base, ST1 + off);
            LDS_READ16(er[ST % DW_RING][1][1], base, ST1 + off + 16);
        }
    }
    template <int Q, int P> __device__ __forceinline__ void issue_w() { LDS_READ_U32(w[P], w2l, P * dw_wrow(CEXP) + 64 * Q); }
    template <int Q> __device__ __forceinline__ void issue_bias() {
        LDS_READ16(acc[Q & 1][0], pb, 64 * Q);
        if constexpr (NSUB == 2) LDS_READ16(acc[Q & 1][1], pb, 64 * Q);
    }
    template <int Q> __device__ __forceinline__ void issue_slope() { LDS_READ16(slope, pb, CEXP * 4 + 64 * Q); }
    template <int PAIR, int N> __device__ __forceinline__ void issue_a3() { LDS_READ16(a3[N & 3], w3l, N * 32 * PW3 + 32 * PAIR); }
    template <int PAIR, int N = 0> __device__ __forceinline__ void issue_a3_first() {
        if constexpr (N < cmin(4, NT16)) { issue_a3<PAIR, N>(); issue_a3_first<PAIR, N + 1>(); }
    }
    __device__ __forceinline__ void preamble() {
        issue_bias<0>();
        issue_slope<0>();
        issue_w<0, 0>(); issue_w<0, 1>(); issue_w<0, 2>();
        preamble_taps<0>();
    }
    template <int ST> __device__ __forceinline__ void preamble_taps() {
        if constexpr (ST < PD) { issue_taps<ST>(); preamble_taps<ST + 1>(); }
    }
    template <int ST, int PAIR, int N = 0> __device__ __forceinline__ void project(f32x4 (&pacc)[NSUB][NT16]) {
        if constexpr (N < NT16) {
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a3[N & 3]) : "n"(dw_outstanding(NT16, NQ, 1, ST, N, NSUB, PD)));
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
        if constexpr (ST < DW_ROWS * NQ) {
            constexpr int q = ST / DW_ROWS, p = ST % DW_ROWS, slot = ST % DW_RING, par = q & 1;
            if constexpr (ST + PD < DW_ROWS * NQ) issue_taps<ST + PD>();
            if constexpr ((q & 1) && p == 0) issue_a3_first<(q >> 1)>();
            // everything issued before taps(ST) has landed with them: w[p], bias (p == 0), slope (p == 2)
            constexpr int OUT = dw_outstanding(NT16, NQ, 0, ST, 0, NSUB, PD);
            if constexpr (NSUB == 2)
                asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(er[slot][0][0]), "+v"(er[slot][0][1]), "+v"(er[slot][1][0]), "+v"(er[slot][1][1]), "+v"(w[p]) : "n"(OUT));
            else asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(er[slot][0][0]), "+v"(er[slot][0][1]), "+v"(w[p]) : "n"(OUT));
            if constexpr (p == 0) {                               // volatile statements keep their order: uses come after the wait
                asm volatile("" : "+v"(acc[par][0]));
                if constexpr (NSUB == 2) asm volatile("" : "+v"(acc[par][1]));
            }
            if constexpr (p == DW_ROWS - 1) asm volatile("" : "+v"(slope));
            const u32x4 du = {w[p] & msk[0], w[p] & msk[1], w[p] & msk[2], w[p] & msk[3]};
            const V8 dg = __builtin_bit_cast(V8, du);
            if constexpr (q + 1 < NQ) {
                __builtin_amdgcn_sched_barrier(0);
                issue_w<q + 1, p>();
            }
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                const V16 e = __builtin_shufflevector(er[slot][s][0], er[slot][s][1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                acc[par][s] = smfma16(dg, e, acc[par][s], idx);
            }
            if constexpr (p == 0 && q + 1 < NQ) {
                __builtin_amdgcn_sched_barrier(0);
                issue_bias<q + 1>();
            }
            if constexpr (p == DW_ROWS - 1) {
#pragma unroll
                for (int s = 0; s < NSUB; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float d = acc[par][s][i];
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
