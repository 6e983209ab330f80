// Residual inverted-residual blocks (stride 1, C -> CEXP -> C, + x) with the activation kept in REGISTERS
// from block to block.  (reference: Residual / Depth_Wise.forward, face_model.py:55-75)
//
// Same three stages per 64-channel slice as kernels_block.hip -- expand GEMM + PReLU into an LDS image,
// 3x3 depthwise conv on the matrix cores, project GEMM accumulated in registers -- with two changes that let a
// block's output feed the next block without leaving the lane that computed it:
//
//  * Pixel tiles are row aligned on BOTH sides.  A tile is 32 lanes' worth of whole image rows (two 16-lane
//    subtiles; a row of 14 uses 14 of its 16 lanes), and wave w owns the same tile as expand input and as
//    project output.  The halo rows a band needs above and below are a tile of their own, expanded by another
//    wave.
//  * The expand GEMM runs on 16x16x32 MFMAs with pixels as columns, like the project GEMM whose result it
//    consumes.  A project lane (pixel m, k-group g) ends up with output channels 16 n + 4 g + i (n < C/16,
//    i < 4).  The expand GEMM's K order is free, so k-step ks, k-group g, element j is DEFINED as channel
//    16 (2 ks + (j >> 2)) + 4 g + (j & 3): exactly what that lane holds.  The expand weights are stored with
//    their columns in that order (pack_rblock), and
//        x_next = round(acc + x + b3)
//    is the whole epilogue: no LDS staging, no shortcut read, no global round trip.  With whole-image
//    workgroups (TR == H) `nblk` blocks run back to back on an image; only the first block reads global
//    memory and only the last writes it.
//
// Every per-slice input (both weight matrices, biases, slopes, the 3x3 weights) arrives as one package by
// LDS-DMA into the buffer the previous slice does not use; the packages of consecutive blocks follow each other
// in memory, so the weight stream simply continues across block boundaries.
//
// A band of TR < H rows (one block per launch) runs the same arithmetic per output value -- the same MFMA
// shapes, operand order and roundings -- so results do not depend on which configuration a batch size selects.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "block_stages.h"

namespace fra {

#ifndef RB_EXP_DEFER
#define RB_EXP_DEFER 0     // 1: expand stage with a tile's PReLU issued behind the MFMAs of the next tile (Exp16::run); same bits,
                           // conv_4 chain at batch 1024 301.3-301.6 us against 300.8-302.4: the partner wave already covers that wait
#endif
#ifndef RB_EXP_PD
#define RB_EXP_PD 3        // expand stage: fragment reads run this many steps ahead of their MFMAs
#endif

// ---- expand stage: E[16-channel tile][pixels] = PReLU(W1 . X + b1) for tiles T0 .. T0 + NTT - 1 of the slice ----
// LDS reads in issue order: bias quads, slope quads, then one A fragment per step S = (tile, k-step), PD steps
// ahead of its MFMAs through a ring of NB registers.  LDS returns in order, so the wait of step S (at most
// `outstanding(S)` reads in flight) also covers every quad.
template <typename T, int KS, int PW1, int T0, int NTT> struct Exp16 {
    typedef typename vec8<T>::type V8;
    static constexpr int NS = NTT * KS, PD = RB_EXP_PD, NB = PD + 1;
    static_assert(NTT % 2 == 0 && NS > PD, "tiles are written in pairs (one 16-byte unit per pair)");
    V8 fr[NB];
    f32x4 bq[NTT], sl[NTT];
    f32x4 acc[2][2];   // [tile parity][subtile]
    V8 o[2];
    int w1l;   // LDS address of this lane's W1 image row 16 (g & 1) + m, unit g >> 1; + 32 t PW1 + 32 ks
    int pl;    // LDS address of the bias quad of channels 4 g ..; + 64 t; slopes 256 further

    template <int S> __device__ __forceinline__ void issue_frag() {
        LDS_READ16(fr[S % NB], w1l, (T0 + S / KS) * 32 * PW1 + (S % KS) * 32);
    }
    template <int I = 0> __device__ __forceinline__ void preamble() {
        if constexpr (I < NTT) { LDS_READ16(bq[I], pl, 64 * (T0 + I)); preamble<I + 1>(); }
        else if constexpr (I < 2 * NTT) { LDS_READ16(sl[I - NTT], pl, 256 + 64 * (T0 + I - NTT)); preamble<I + 1>(); }
        else if constexpr (I < 2 * NTT + PD) { issue_frag<I - 2 * NTT>(); preamble<I + 1>(); }
    }
    static constexpr int outstanding(int S) { return cmin(PD, NS - 1 - S); }
    // PReLU of values [I0, I1) of tile t (both subtiles) into the output unit of the tile pair: sl = slope - 1
    template <int t, int I0, int I1> __device__ __forceinline__ void prelu_piece() {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int i = I0; i < I1; ++i) {
                const float v = acc[t & 1][st][i];
                o[st][4 * (t & 1) + i] = (T)fmaf(neg_part(v), sl[t][i], v);
            }
    }
    // positions 16 g + 4 t + i of tiles t = T0 + tl - 1, T0 + tl (tl odd): one 16-byte unit per subtile
    template <int tl> __device__ __forceinline__ void write_pair(char* lds, const int (&lds_px)[2]) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
            if (lds_px[st] >= 0) *reinterpret_cast<V8*>(lds + lds_px[st] + 8 * (T0 + tl - 1)) = o[st];
    }
    // lds_px[st]: LDS byte address of this lane's 32-byte share of its pixel slot in subtile st (-1: no pixel)
    // The PReLU of a tile is issued one tile LATER, a piece behind each MFMA pair of the next tile: its operands have long
    // left the matrix pipe by then (straight behind the tile's last MFMA it would wait out that instruction's latency).
    template <int S = 0> __device__ __forceinline__ void run(const V8 (&x)[2][KS], char* lds, const int (&lds_px)[2]) {
        static_assert(4 % KS == 0, "a tile's four values per lane are dealt over its k-steps");
        if constexpr (S < NS) {
            constexpr int tl = S / KS, ks = S % KS, slot = S % NB, a = tl & 1;
            if constexpr (S + PD < NS) issue_frag<S + PD>();
            if constexpr (ks == 0)
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fr[slot]), "+v"(bq[tl]) : "n"(outstanding(S)));
            else if constexpr (ks == KS - 1)
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fr[slot]), "+v"(sl[tl]) : "n"(outstanding(S)));
            else
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fr[slot]) : "n"(outstanding(S)));
            if constexpr (ks == 0) {
                acc[a][0] = mfma16(fr[slot], x[0][0], bq[tl]);
                acc[a][1] = mfma16(fr[slot], x[1][0], bq[tl]);
            } else {
                acc[a][0] = mfma16(fr[slot], x[0][ks], acc[a][0]);
                acc[a][1] = mfma16(fr[slot], x[1][ks], acc[a][1]);
            }
            __builtin_amdgcn_sched_barrier(0);             // keep each MFMA pair right behind its wait
#if RB_EXP_DEFER
            if constexpr (tl > 0) {
                prelu_piece<tl - 1, ks * (4 / KS), (ks + 1) * (4 / KS)>();
                if constexpr (ks == KS - 1 && ((tl - 1) & 1)) write_pair<tl - 1>(lds, lds_px);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (S == NS - 1) {                   // the last tile has no successor to hide behind
                prelu_piece<tl, 0, 4>();
                if constexpr (tl & 1) write_pair<tl>(lds, lds_px);
                __builtin_amdgcn_sched_barrier(0);
            }
#else
            if constexpr (ks == KS - 1) {
                prelu_piece<tl, 0, 4>();
                if constexpr (tl & 1) write_pair<tl>(lds, lds_px);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            run<S + 1>(x, lds, lds_px);
        }
    }
};

// N quads, 64 bytes apart, read from LDS behind one wait
template <int N> struct QuadLoad {
    f32x4 q[N];
    template <int I = 0> __device__ __forceinline__ void issue(int addr) {
        if constexpr (I < N) { LDS_READ16(q[I], addr, 64 * I); issue<I + 1>(addr); }
    }
    // volatile statements keep their order: placed after the wait, whatever uses a quad comes after the wait too
    template <int I = 0> __device__ __forceinline__ void pin() {
        if constexpr (I < N) { asm volatile("" : "+v"(q[I])); pin<I + 1>(); }
    }
};

// -DRB_STAMPS: per wave, shader-clock cycles per part of the slice loop, summed over the kernel (diagnostic builds only)
#ifdef RB_STAMPS
__device__ unsigned long long fra_rb_cycles[65536 * 8 * 8];
extern "C" int fra_dbg_read(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fra_rb_cycles), bytes, 0, hipMemcpyDeviceToHost);
}
#define RB_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define RB_ACC(i, a, b) dbg[i] += (b) - (a)
#else
#define RB_T(var)
#define RB_ACC(i, a, b)
#endif

#ifndef RB_FLAGS
#define RB_FLAGS 1         // whole-image workgroups synchronise wave to wave through flags in LDS instead of barriers
#endif
#ifndef RB_OPPOSED
#define RB_OPPOSED 0       // 1: whole-image workgroups synchronise once per ROUND (depthwise stage of slice r + expand stage of slice
                           // r + 1, the two waves of a SIMD in opposite order).  Same bits; measured at batch 1024, six blocks of
                           // conv_4, one box: 322-324 us against 312-315 us for the slice loop (every wave depthwise-first: 316)
#endif
#ifndef RB_EXPAND_FIRST_WAVES
#define RB_EXPAND_FIRST_WAVES 0x0F   // bit w: wave w runs a round's expand stage before its depthwise stage (the first wave of each SIMD)
#endif
// -DRB_PRIO_POLICY=n (experiments): issue priority of the two waves of a SIMD by stage.  old = dispatched first (waves 0-3)
//   1: young waves 1 throughout, old waves 1 in the expand stage and 0 in the depthwise stage;  2: the reverse for the old waves;
//   3: every wave 1 in its depthwise stage, 0 in its expand stage;  4: old waves as 3, young waves the reverse
//   6-8: as 3 with the young (6, 7) or the old (8) waves one level higher.   Batch 1024, six blocks of conv_4, one box, us per
//   launch: 0: 307.4  1: 303.3  2: 301.8  3: 300.7-302.0  4: 309.0  6: 304.6  7: 301.0  8: 302.4  (young waves 1 always: 308.5)
#ifndef RB_PRIO_POLICY
#define RB_PRIO_POLICY 3
#endif
#if RB_PRIO_POLICY == 0
#define RB_PRIO_STAGE(stage)
#else
#define RB_PRIO_STAGE(stage)                                                                              \
    do {                                                                                                  \
        const bool old_ = wave < NW / 2;                                                                  \
        int lvl_;                                                                                         \
        if (RB_PRIO_POLICY == 1) lvl_ = !old_ || (stage) == 0;                                            \
        else if (RB_PRIO_POLICY == 2) lvl_ = !old_ || (stage) == 1;                                       \
        else if (RB_PRIO_POLICY == 3) lvl_ = (stage) == 1;                                                \
        else if (RB_PRIO_POLICY == 4) lvl_ = old_ ? (stage) == 1 : (stage) == 0;                          \
        else if (RB_PRIO_POLICY == 6) lvl_ = ((stage) == 1 ? 1 : 0) + (old_ ? 0 : 1);                     \
        else if (RB_PRIO_POLICY == 7) lvl_ = (stage) == 1 ? (old_ ? 1 : 2) : 0;                           \
        else lvl_ = (stage) == 1 ? (old_ ? 2 : 1) : 0;                                                    \
        if (lvl_ == 0) __builtin_amdgcn_s_setprio(0);                                                     \
        else if (lvl_ == 1) __builtin_amdgcn_s_setprio(1);                                                \
        else __builtin_amdgcn_s_setprio(2);                                                               \
    } while (0)
#endif
#ifndef RB_DS_DW_PD
#define RB_DS_DW_PD 1      // dual-slice band launches: steps the depthwise stage's tap reads run ahead (2: pair launch 16.1 us against 15.9)
#endif
#ifndef RB_SPIN_LIMIT
#define RB_SPIN_LIMIT (1 << 20)   // polls (some 0.1 s) after which a wait gives up and reports it through `fault` (common.h)
#endif
// whole-image workgroups keep TWO activation images (slices alternate) and a block of flag words
//
// G > 1 (7 x 7 maps): a workgroup owns G whole images at a time, STACKED so that its waves have tiles to work on.  An
// image is 7 rows of 8 slots (7 pixels and a border slot that is the next row's left border as well), images are 64
// slots apart: the 9 slots between two images are the lower border row of the one and the upper border row of the
// other.  The two activation image sets are G * 64 slots apart for the same reason; 16 slots follow the second one
// (its last border row, and the reads of lanes that have no pixel).
// CH > 1 (bands of small batches): CH blocks per launch.  A band's workgroup computes the first block on CH - 1 more rows
// above and below (from CH more input rows each side), the next on one row less each side, ...: no row ever comes from
// another workgroup, at the price of computing the rows next to a band twice -- which costs a small batch nothing, while
// every launch it saves is some 9 microseconds.
// DUAL SLICES (CH > 1, DUAL; small_batch_mode() == 1, the default, selects it).  A band workgroup of a small batch has three input tiles and two output tiles for eight waves, and
// a slice is a chain of latencies -- package, expand stage, barrier, depthwise + project stage, barrier -- that its little
// work does not fill.  So slices 2 r and 2 r + 1 of a block run in the same round: every expand wave expands its
// channel tiles of BOTH slices into two activation images, then the depthwise + project work of slice 2 r goes to the
// waves of tiles 0 and 1 and that of slice 2 r + 1 to the waves that would otherwise wait (tiles 2 and 3 of the expand
// side), each with project sums of its own.  At the end of a block the second set of sums is handed over through LDS and
// added to the first: x <- round(((even slices + odd slices) + x) + b3) -- the project sum is associated differently from
// the one-slice-at-a-time kernels, so batches that run here (7 B <= compute units) agree with larger batches to rounding,
// not bit for bit; among themselves they agree bit for bit.  The two groups of a package (RbPkg) have buffers of their own
// per slice of the round: the expand groups of round r + 1 arrive during the depthwise stage of round r, the depthwise /
// project groups of round r during its expand stage.
template <int H, int TR, int G, int CH = 1> constexpr int rb_es_bytes() {
    return G > 1 ? G * 64 * 144 : (TR + 2 * CH) * block_pitch<H, 1>() * 144;
}
// main tiles (expand input == project output) of a workgroup: whole rows, padded to 8, 16 or 32 lanes, 32 lanes per tile
template <int H, int TR, int G> constexpr int rb_main_tiles() {
    constexpr int rpt = 32 / (H <= 8 ? 8 : (H <= 16 ? 16 : 32));
    return G > 1 ? (G / 2 * H + 1) / 2 : (TR + rpt - 1) / rpt;
}
// flags instead of barriers: whole images, one tile per wave (no two waves sharing a tile) and a wave to spare
template <int H, int TR, int NW, int G> constexpr bool rb_flags() {
    return TR == H && RB_FLAGS && rb_main_tiles<H, TR, G>() > NW / 2 && rb_main_tiles<H, TR, G>() < NW;
}
template <int H, int TR, int NW, int G, int CH = 1, bool DUAL = false> constexpr int rb_images_bytes() {
    return G > 1 ? (2 * rb_es_bytes<H, TR, G>() + 16 * 144 + 1023) / 1024 * 1024
                 : (rb_flags<H, TR, NW, G>() || DUAL ? 2 : 1) * rb_es_bytes<H, TR, G, CH>();
}
template <int C, int CEXP, int H, int TR, int NW, int G = 1, int CH = 1, bool DUAL = false> constexpr int rblock_lds_bytes() {
    return rb_images_bytes<H, TR, NW, G, CH, DUAL>() + 2 * RbPkg<C>::BYTES + (rb_flags<H, TR, NW, G>() ? 128 : 0) +
           (CH > 1 ? (rb_main_tiles<H, TR, G>() + CH - 1) * 2 * 64 * (DUAL ? C / 16 : C / 32) * 16 : 0);
}

template <typename T, int C, int CEXP, int H, int TR, int NW, int G, int CH, bool DUAL>
__device__ __forceinline__ void rblock_body(const T* __restrict__ X, const char* __restrict__ WP, T* __restrict__ O, int B, int nblk,
                                            unsigned* __restrict__ fault, int spin_limit);

template <typename T, int C, int CEXP, int H, int TR, int NW, int G = 1, int CH = 1>
__global__ __launch_bounds__(NW * 64, 2)
void rblock_kernel(const T* __restrict__ X, const char* __restrict__ WP, T* __restrict__ O, int B, int nblk,
                   unsigned* __restrict__ fault, int spin_limit) {
    rblock_body<T, C, CEXP, H, TR, NW, G, CH, false>(X, WP, O, B, nblk, fault, spin_limit);
}

// the same body with the register budget of exactly two waves per SIMD spelled out (the several-blocks-per-band
// configuration: left to itself the compiler aims at three waves and spills)
template <typename T, int C, int CEXP, int H, int TR, int NW, int G, int CH, bool DUAL>
__global__ __launch_bounds__(NW * 64, 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
void rblock_kernel_w2(const T* __restrict__ X, const char* __restrict__ WP, T* __restrict__ O, int B, int nblk,
                      unsigned* __restrict__ fault, int spin_limit) {
    rblock_body<T, C, CEXP, H, TR, NW, G, CH, DUAL>(X, WP, O, B, nblk, fault, spin_limit);
}

template <typename T, int C, int CEXP, int H, int TR, int NW, int G, int CH, bool DUAL>
__device__ __forceinline__ void rblock_body(const T* __restrict__ X, const char* __restrict__ WP, T* __restrict__ O, int B, int nblk,
                                            unsigned* __restrict__ fault, int spin_limit) {
    typedef typename vec8<T>::type V8;
    typedef typename vec4<T>::type V4;
    typedef RbPkg<C> PK;
    constexpr int W = H;
    constexpr bool WHOLE = (TR == H);                      // the band is the whole image: no halo rows
    constexpr int RIN = TR + 2 * CH;                       // LDS rows: band rows + CH rows above and below
    // a tile is RPT whole rows, each padded to POW (a power of two) lanes
    constexpr int POW = W <= 8 ? 8 : (W <= 16 ? 16 : 32), RPT = 32 / POW;
    // STACK: subtile st of every tile belongs to images st GP .. st GP + GP - 1 of the workgroup's G, tile t to rows
    // 2 t, 2 t + 1 of those GP images' 7 GP rows (two rows of 7 pixels per 16-lane subtile)
    constexpr bool STACK = G > 1;
    constexpr int GP = STACK ? G / 2 : 1;
    constexpr int MTO = rb_main_tiles<H, TR, G>();         // main tiles: expand input == project output
    static_assert(MTO == (STACK ? (GP * H + 1) / 2 : (TR + RPT - 1) / RPT), "tiles of whole rows");
    // halo tiles: tile MTO + j holds the rows y0 - 1 - j and y0 + TR + j (RPT == 1: a tile each for y0 - 1 and y0 + TR)
    constexpr int HT = WHOLE ? 0 : (RPT >= 2 ? CH : 2);
    static_assert(CH == 1 || (!WHOLE && G == 1 && RPT >= 2), "several blocks per band launch: halo rows share a tile");
    constexpr int MTI = MTO + HT;
    constexpr int KS = C / 32, NT16 = C / 16, NCH = CEXP / 64;
    constexpr int PITCH = STACK ? W + 1 : block_pitch<H, 1>(), SLOT = 144;
    constexpr int ES_BYTES = rb_es_bytes<H, TR, G, CH>();
    static_assert(STACK || ES_BYTES == RIN * PITCH * SLOT, "band rows and their borders");
    // FL: no workgroup-wide barriers in the slice loop.  A wave's expand stage writes its tile's rows, its depthwise
    // stage reads those and the two neighbouring tiles' rows: it waits for those two waves only (flags in LDS), and
    // with two activation images (slice parity) a wave may expand the next slice while its neighbours still read this
    // one.  The tile-less last wave copies the weight packages and tells the others when one has landed.
    constexpr bool FL = rb_flags<H, TR, NW, G>();
    constexpr int NIMG = FL ? 2 : 1;
    constexpr int IMG_BYTES = rb_images_bytes<H, TR, NW, G, CH, DUAL>();
    constexpr int OFF_PKG = IMG_BYTES, OFF_FLAGS = OFF_PKG + 2 * PK::BYTES;
    static_assert(!STACK || (FL && H == 7 && (G == 2 || G == 4)), "stacked images: 7 x 7 maps, whole-image workgroups with flags");
    static_assert(IMG_BYTES >= NIMG * ES_BYTES, "both activation images fit");
    constexpr int OFF_XCH = OFF_FLAGS + (FL ? 128 : 0);
    constexpr bool DS = DUAL;
    static_assert(!DUAL || CH > 1, "dual slices are a mode of the several-blocks-per-band launches");
    // per output tile and subtile: a lane's KS fragments (DS: first the NT16 quads of the odd slices' project sums)
    constexpr int XCH_SLOT = 64 * (DS ? NT16 : KS) * 16;
    constexpr int XCH_BYTES = CH > 1 ? (MTO + CH - 1) * 2 * XCH_SLOT : 0;
    constexpr int LDS_BYTES = OFF_XCH + XCH_BYTES;
    constexpr int MTIW = (MTI + NW - 1) / NW;              // input tiles per wave
    // with at most NW / 2 input tiles two waves share one and expand two of the slice's four channel tiles each
    constexpr bool ESPLIT = MTI <= NW / 2;
    constexpr int MTIP = MTI > 2 ? 4 : (MTI > 1 ? 2 : 1);
    static_assert(CEXP % 128 == 0 && C % 32 == 0 && H % TR == 0, "shape");
    static_assert(MTO <= NW, "one output tile per wave");
    // with at most NW / 4 main tiles the two waves that share an input tile also share its depthwise / project work:
    // one 16-pixel subtile each (pixels are columns of every product, so nothing about a value changes)
    constexpr bool SSPLIT = ESPLIT && 4 * (MTO + CH - 1) <= NW;
    // (CH > 1) the two waves of a tile hand each other their subtile of a block's output through LDS (OFF_XCH)
    static_assert(CH == 1 || SSPLIT, "several blocks per band launch: two waves per tile, a subtile each");
    constexpr int NSUB = SSPLIT ? 1 : 2;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(LDS_BYTES == rblock_lds_bytes<C, CEXP, H, TR, NW, G, CH, DUAL>(), "the launcher's LDS size is this map's");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15, g4 = lane >> 4;
    const int y0 = blockIdx.x * TR;                        // first image row of the band
    RB_T(t_entry);
    const int npk = nblk * NCH;                            // packages of this launch

    // ---- package staging: piece k (1 KB) is copied by wave k % NW ----------------------------------------------
    auto stage = [&](int p, int buf) {
        const char* g = WP + (size_t)p * PK::BYTES + lane * 16;
        char* l = lds + OFF_PKG + buf * PK::BYTES;
#pragma unroll
        for (int j = 0; j < (PK::PIECES + NW - 1) / NW; ++j) {
            const int k = wave + NW * j;
            if (k < PK::PIECES)                            // wave-uniform
                __builtin_amdgcn_global_load_lds((GPTR)(g + k * 1024), (LPTR)(l + k * 1024), 16, 0, 0);
        }
    };
    if constexpr (!FL && !DS) stage(0, 0);
#ifdef RB_PRIO_YOUNG
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses issue arbitration otherwise
#endif

    // the activation image starts as zeros: border slots and out-of-image rows are never written afterwards
    for (int i = tid; i < IMG_BYTES / 16; i += NW * 64) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // flags (FL): exp_done[w] / dw_done[w] = slices wave w has expanded / finished reading, pkg_ready = packages landed
    volatile int* exp_done = reinterpret_cast<volatile int*>(lds + OFF_FLAGS);
    volatile int* dw_done = exp_done + 8;
    volatile int* pkg_ready = exp_done + 16;
    if (FL && tid < 32) exp_done[tid] = 0;

    // ---- this lane's pixels: tile ti, subtile st -> LDS slot and offset inside an image -------------------------
    const int we = ESPLIT ? (wave & (MTIP - 1)) : wave, en = ESPLIT ? wave / MTIP : 0;   // wave-uniform
    int lds_px[MTIW][2];   // LDS byte address of the lane's 32 bytes (positions 16 g ..) of the pixel slot; -1 = no pixel
    int x_off[MTIW][2];    // element offset of the lane's first channel (4 g) of that pixel inside an image (STACK: inside the G images)
    int img_of[MTIW][2];   // (STACK) which of the workgroup's G images the pixel belongs to
#pragma unroll
    for (int tt = 0; tt < MTIW; ++tt)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int ti = we + NW * tt;
            img_of[tt][st] = 0;
            if constexpr (STACK) {
                const int pr = 2 * ti + (m16 >> 3), xx = m16 & 7;      // row among the 7 GP rows of the subtile's images
                const int gi = st * GP + pr / H, r = pr % H;
                const bool ok = ti < MTO && pr < GP * H && xx < W;
                lds_px[tt][st] = ok ? (gi * 64 + PITCH + 1 + r * PITCH + xx) * SLOT + 32 * g4 : -1;
                x_off[tt][st] = ok ? ((gi * H + r) * W + xx) * C + 4 * g4 : 4 * g4;
                if (ok) img_of[tt][st] = gi;
                continue;
            }
            int lr, xx, gy;
            bool ok;
            const int rr = POW == 32 ? 0 : (POW == 16 ? st : 2 * st + (m16 >> 3));
            xx = POW == 32 ? 16 * st + m16 : (POW == 16 ? m16 : (m16 & 7));
            if (ti < MTO) {
                const int orow = ti * RPT + rr;
                ok = orow < TR;
                lr = CH + orow;
                gy = y0 + orow;
            } else {
                const int which = POW == 32 ? ti - MTO : rr;   // 0: a row above the band, 1: a row below
                const int lvl = POW == 32 ? 0 : ti - MTO;      // how far from the band
                ok = ti < MTI && which < 2;
                lr = which ? CH + TR + lvl : CH - 1 - lvl;
                gy = which ? y0 + TR + lvl : y0 - 1 - lvl;
            }
            ok = ok && xx < W && gy >= 0 && gy < H;
            lds_px[tt][st] = ok ? (lr * PITCH + xx + 1) * SLOT + 32 * g4 : -1;
            x_off[tt][st] = ok ? (gy * W + xx) * C + 4 * g4 : 4 * g4;
        }

    // input fragments = B operand of the expand GEMM = the shortcut; element j of k-step ks is channel
    // 32 ks + 4 g + j (j < 4) or 32 ks + 16 + 4 g + (j - 4)
    V8 xf[MTIW][2][KS];
    // (STACK: `img` counts groups of G images; a lane whose image lies beyond the batch reads its pixel of the group's first one)
    auto load_x = [&](int img) {
        const T* Ximg = X + (size_t)img * G * H * W * C;
#pragma unroll
        for (int tt = 0; tt < MTIW; ++tt)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                int off = x_off[tt][st];
                if constexpr (STACK)
                    if (img * G + img_of[tt][st] >= B) off -= img_of[tt][st] * H * W * C;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const V4 lo = *reinterpret_cast<const V4*>(Ximg + off + 32 * ks);
                    const V4 hi = *reinterpret_cast<const V4*>(Ximg + off + 32 * ks + 16);
                    xf[tt][st][ks] = V8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
    };

    // ---- depthwise / project side: wave wt owns output tile wt (its own main input tile), or one subtile of it --
    // (DS) tiles 0 and 1 are output tiles; the waves of expand tiles 2 and 3 take the same two tiles for the round's odd slice
    static_assert(!DS || (MTO == 1 && CH == 2 && MTIP == 4 && NW == 8 && NCH % 2 == 0), "dual slices: one main tile, two blocks, eight waves");
    const int wt = DS ? (we & 1) : (SSPLIT ? we : wave);
    const int dpar = DS ? (we >> 1) : 0;                   // (DS) which slice of a round this wave's depthwise + project stage works on
    const bool dw_active = SSPLIT ? (we < MTO && en < 2) : (wt < MTO);
    constexpr int ST1 = STACK ? GP * 64 * SLOT : (POW == 32 ? 16 * SLOT : (16 / POW) * PITCH * SLOT);
    static_assert(ST1 + 2 * PITCH * SLOT + 112 < 65536, "tap offsets must fit the instruction's offset field");
    int tb;               // LDS byte address of this lane's tap (0, min(g, 2)) (DwStage::tb)
    {
        const int oy = wt * RPT + m16 / POW, ox = m16 % POW;
        const int sub = SSPLIT ? en * ST1 : 0;                                 // (SSPLIT) this wave's subtile
        tb = ((CH - 1 + oy) * PITCH + ox + (g4 < 2 ? g4 : 2)) * SLOT + sub;
        // (CH > 1) the first halo tile as an OUTPUT tile: subtile 0 is row y0 - 1 (LDS row CH - 1), subtile 1 row y0 + TR
        // (14-wide maps: a subtile is a row; 7-wide maps: subtile 0 holds both rows, 8 lanes each, subtile 1 no pixel)
        if (CH > 1 && wt == MTO) {
            const int below = POW == 16 ? en : (m16 >> 3);
            tb = ((CH - 2 + below * (TR + 1)) * PITCH + ox + (g4 < 2 ? g4 : 2)) * SLOT;
        }
        if constexpr (STACK) {                             // pixel slot less one row and one slot
            const int pr = 2 * wt + (m16 >> 3);
            tb = ((pr / H) * 64 + (pr % H) * PITCH + (m16 & 7) + (g4 < 2 ? g4 : 2)) * SLOT;
        }
    }
    f32x4 pacc[NSUB][NT16];  // tile n of subtile st, register i <-> output channel 16 n + 4 g + i

    // blk: (CH > 1) block of the launch being computed: its tiles shrink by one halo tile per block
    // base: LDS address of the package whose expand group is read (groups that sit in separate buffers: that group's address)
    auto expand = [&](int base, int ioff, int blk = 0) {   // ioff: byte offset of the activation image to write
        if (we >= MTI - blk || en >= 2) return;            // wave-uniform
        const int w1l = base + (16 * (g4 & 1) + m16) * PK::PW1 + 16 * (g4 >> 1);
        const int pl = base + PK::OFF_B1 + 16 * g4;
#pragma unroll
        for (int tt = 0; tt < MTIW; ++tt) {
            if (we + NW * tt >= MTI) break;                // wave-uniform
            if constexpr (ESPLIT) {
                if (en == 0) { Exp16<T, KS, PK::PW1, 0, 2> ex; ex.w1l = w1l; ex.pl = pl; ex.preamble(); ex.run(xf[tt], lds + ioff, lds_px[tt]); }
                else         { Exp16<T, KS, PK::PW1, 2, 2> ex; ex.w1l = w1l; ex.pl = pl; ex.preamble(); ex.run(xf[tt], lds + ioff, lds_px[tt]); }
            } else {
                Exp16<T, KS, PK::PW1, 0, 4> ex; ex.w1l = w1l; ex.pl = pl; ex.preamble(); ex.run(xf[tt], lds + ioff, lds_px[tt]);
            }
        }
    };
    // base: LDS address the package would start at for its depthwise / project group to sit where it does
    auto dwproject = [&](int base, int ioff, int blk = 0) {
        if (CH == 1 ? !dw_active : (wt >= MTO + CH - 1 - blk || en >= 2)) return;   // wave-uniform
        DwStage<T, 64, NT16, PK::PW3, ST1, PITCH * SLOT, 4, NSUB, false, (DS ? RB_DS_DW_PD : DW_PD)> dw;
        dw.tb = tb + ioff;
        dw.set_lane(m16, g4);
        dw.w2l = base + PK::OFF_W2 + (g4 & 1) * dw_wplane(64) + m16 * 4;
        dw.pb = base + PK::OFF_B2 + 16 * g4;
        dw.w3l = base + PK::OFF_W3 + (16 * (g4 & 1) + m16) * PK::PW3 + 16 * (g4 >> 1);
        dw.preamble();
        dw.run(pacc);
    };

#ifdef RB_STAMPS
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    RB_T(t_begin);
    RB_ACC(6, t_entry, t_begin);                           // set-up before the loops: zeroed images, lane constants
#endif
    // ---- epilogue of a block in registers: x <- round((acc + x) + b3), channel for channel what the lane holds; the last
    // block of a launch also stores the image ----
    auto block_epilogue = [&](int base, bool store, int img) {
        QuadLoad<NT16> bq3;                                // b3 of channels 16 n + 4 g ..
        bq3.issue(base + PK::OFF_B3 + 16 * g4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bq3.pin();
        // subtile st of the tile, accumulated in pacc[a]
        auto finish = [&](auto st_c, auto a_c) {
            constexpr int st = decltype(st_c)::value, a = decltype(a_c)::value;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = 2 * ks + (j >> 2), i = j & 3;
                    const float v = pacc[a][n][i] + (float)xf[0][st][ks][j];
                    xf[0][st][ks][j] = (T)(v + bq3.q[n][i]);
                }
            if (store && lds_px[0][st] >= 0 && (!STACK || img * G + img_of[0][st] < B)) {
                T* Oimg = O + (size_t)img * G * H * W * C;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const V8 v = xf[0][st][ks];
                    *reinterpret_cast<V4*>(Oimg + x_off[0][st] + 32 * ks) = V4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<V4*>(Oimg + x_off[0][st] + 32 * ks + 16) = V4{v[4], v[5], v[6], v[7]};
                }
            }
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 1> I1;
        if constexpr (SSPLIT) {
            if (en == 0) finish(I0{}, I0{});               // wave-uniform
            else finish(I1{}, I0{});
        } else {
            finish(I0{}, I0{});
            finish(I1{}, std::integral_constant<int, NSUB - 1>{});
        }
    };

    if constexpr (FL) {
        // ================= whole-image workgroups: wave-to-wave flags, no barrier in the slice loop =================
        // Slices are numbered through (s = 0, 1, ... over images, blocks and slices); slice s uses activation image and
        // package buffer s & 1.
        static_assert(MTIW == 1 && !ESPLIT && MTI == MTO && MTO < NW, "one tile per wave and a wave to spare");
        __syncthreads();                                   // zeroed images and flags
        const int NU = (B + G - 1) / G;                    // images, or groups of G stacked images
        const int nimg = (NU - (int)blockIdx.y + (int)gridDim.y - 1) / (int)gridDim.y;
        const int NS = nimg * npk;
        // spin until `cond` holds for every lane (flags only ever grow)
        auto wait_flags = [&](auto cond) {
            for (int spin = 0;; ++spin) {
                if (__builtin_amdgcn_ballot_w64(!cond()) == 0) break;
                if (spin >= spin_limit) {
                    if (fault && lane == 0) *fault = 0x52420000u | (unsigned)wave;   // 'RB'
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
        };
#if RB_OPPOSED
        // ROUNDS.  Round r of a wave is { depthwise + project stage of slice r, expand stage of slice r + 1 }: with two
        // activation images (slice parity) and the package's two groups in buffers of their own, both read and write
        // only what round r - 1 left behind, so a wave synchronises with its neighbours ONCE per round, and the order of
        // the two inside a round is free.  The waves dispatched first (one per SIMD) expand first, the others run the
        // depthwise stage first: the two waves of a SIMD are then in opposite stages -- matrix-heavy beside LDS- and
        // vector-heavy -- instead of competing for the same pipe.  At the end of a block the order is fixed (the next
        // block's expand stage needs this block's output).
        volatile int* e_ready = exp_done + 16;             // expand groups landed
        volatile int* d_ready = exp_done + 17;             // depthwise / project groups landed
        constexpr int OFF_EG = OFF_PKG, OFF_DG = OFF_PKG + 2 * PK::EG_BYTES;
        auto e_base = [&](int q) { return OFF_EG + (q & 1) * PK::EG_BYTES; };
        auto d_base = [&](int q) { return OFF_DG + (q & 1) * PK::DG_BYTES - PK::EG_BYTES; };
        if (wave == NW - 1) {
            auto load_group = [&](int q, bool dgroup) {    // slice q's expand or depthwise / project group
                const char* g = WP + (size_t)(q % npk) * PK::BYTES + (dgroup ? PK::EG_BYTES : 0) + lane * 16;
                char* l = lds + (dgroup ? OFF_DG + (q & 1) * PK::DG_BYTES : OFF_EG + (q & 1) * PK::EG_BYTES);
                if (dgroup) {
#pragma unroll
                    for (int k = 0; k < PK::DG_PIECES; ++k)
                        __builtin_amdgcn_global_load_lds((GPTR)(g + k * 1024), (LPTR)(l + k * 1024), 16, 0, 0);
                } else {
#pragma unroll
                    for (int k = 0; k < PK::EG_PIECES; ++k)
                        __builtin_amdgcn_global_load_lds((GPTR)(g + k * 1024), (LPTR)(l + k * 1024), 16, 0, 0);
                }
            };
            load_group(0, false);
            if (NS > 1) load_group(1, false);
            load_group(0, true);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) { *e_ready = NS > 1 ? 2 : 1; *d_ready = 1; }
            for (int r = 0; r < NS; ++r) {
                // every wave has finished round r - 1: the expand group of slice r and the other group of slice r - 1 are free
                wait_flags([&] { return lane >= MTO || (exp_done[lane & 7] >= r + 1 && dw_done[lane & 7] >= r); });
                if (r + 2 < NS) load_group(r + 2, false);
                if (r + 1 < NS) load_group(r + 1, true);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) { *e_ready = cmin_rt(r + 3, NS); *d_ready = cmin_rt(r + 2, NS); }
            }
        } else if (wave < MTO) {
            const int nb = (lane & 1) ? wave + 1 : wave - 1;
            const bool has_nb = nb >= 0 && nb < MTO;
            const int nbc = has_nb ? nb : wave;
            const bool expand_first = ((RB_EXPAND_FIRST_WAVES) >> wave) & 1;   // wave-uniform
            auto do_expand = [&](int q) {                  // slice q: its image's first slice loads the image
                if (q % npk == 0) load_x((int)blockIdx.y + (q / npk) * (int)gridDim.y);
                RB_T(ta);
                expand(e_base(q), (q & 1) * ES_BYTES);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the rows are in LDS before the flag goes up
                if (lane == 0) exp_done[wave] = q + 1;
                RB_T(tb_);
                RB_ACC(0, ta, tb_);
            };
#pragma unroll 1
            for (int r = -1; r < NS; ++r) {
                const bool has_next = r + 1 < NS;
                const int c = r < 0 ? NCH - 1 : r % NCH;
                const bool block_end = c == NCH - 1;
                RB_T(t0);
                // what round r - 1 left: the neighbours' rows of slice r, their reads of slice r - 1's rows; this round's groups
                wait_flags([&] {
                    return *e_ready >= cmin_rt(r + 2, NS) && *d_ready >= r + 1 &&
                           (!has_nb || (exp_done[nbc] >= r + 1 && dw_done[nbc] >= r));
                });
                RB_T(t1);
                RB_ACC(1, t0, t1);
                const bool first = expand_first && !block_end && has_next;
                if (first) do_expand(r + 1);
                if (r >= 0) {
                    if (c == 0) {
#pragma unroll
                        for (int st = 0; st < NSUB; ++st)
#pragma unroll
                            for (int n = 0; n < NT16; ++n) pacc[st][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    RB_T(t3);
                    dwproject(d_base(r), (r & 1) * ES_BYTES);
                    RB_T(t4);
                    RB_ACC(2, t3, t4);
                    if (block_end) block_epilogue(d_base(r), (r / NCH) % nblk == nblk - 1, (int)blockIdx.y + (r / npk) * (int)gridDim.y);
                    asm volatile("" ::: "memory");
                    if (lane == 0) dw_done[wave] = r + 1;                      // every read of image and package has returned
                    RB_T(t5);
                    RB_ACC(4, t4, t5);
                }
                if (has_next && !first) do_expand(r + 1);
            }
        }
#else
        if (wave == NW - 1) {
            // ---- loader: package s goes into buffer s & 1 once every wave has finished reading slice s - 2 ----
            // (loading the half the expand stage reads as soon as every wave has expanded slice s - 2, with a flag of
            // its own, was measured: no faster)
            for (int s = 0; s < NS; ++s) {
                if (s >= 2) wait_flags([&] { return lane >= MTO || dw_done[lane & 7] >= s - 1; });
                const char* g = WP + (size_t)(s % npk) * PK::BYTES + lane * 16;
                char* l = lds + OFF_PKG + (s & 1) * PK::BYTES;
#pragma unroll
                for (int k = 0; k < PK::PIECES; ++k)
                    __builtin_amdgcn_global_load_lds((GPTR)(g + k * 1024), (LPTR)(l + k * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) *pkg_ready = s + 1;
            }
        } else if (wave < MTO) {
            // neighbours whose rows this wave's depthwise window touches (and who read this wave's rows)
            const int nb = (lane & 1) ? wave + 1 : wave - 1;
            const bool has_nb = nb >= 0 && nb < MTO;
            const int nbc = has_nb ? nb : wave;
            int s = 0;
            for (int img = blockIdx.y; img < NU; img += gridDim.y) {
                load_x(img);
#pragma unroll 1
                for (int b = 0; b < nblk; ++b) {
#pragma unroll
                    for (int st = 0; st < NSUB; ++st)
#pragma unroll
                        for (int n = 0; n < NT16; ++n) pacc[st][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                    for (int c = 0; c < NCH; ++c, ++s) {
                        const int buf = s & 1, ioff = buf * ES_BYTES;
                        RB_T(t0);
                        // package s has landed; the neighbours have finished reading this image's previous contents (slice s - 2)
                        wait_flags([&] { return *pkg_ready >= s + 1 && (!has_nb || dw_done[nbc] >= s - 1); });
                        RB_T(t1);
                        RB_ACC(1, t0, t1);
                        RB_PRIO_STAGE(0);
                        expand(OFF_PKG + buf * PK::BYTES, ioff);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the rows are in LDS before the flag goes up
                        if (lane == 0) exp_done[wave] = s + 1;
                        RB_T(t2);
                        RB_ACC(0, t1, t2);
                        wait_flags([&] { return !has_nb || exp_done[nbc] >= s + 1; });
                        RB_T(t3);
                        RB_ACC(3, t2, t3);
                        RB_PRIO_STAGE(1);
                        dwproject(OFF_PKG + buf * PK::BYTES, ioff);
                        RB_T(t4);
                        RB_ACC(2, t3, t4);
                        if (c == NCH - 1) block_epilogue(OFF_PKG + buf * PK::BYTES, b == nblk - 1, img);   // reads b3 from the package
                        asm volatile("" ::: "memory");
                        if (lane == 0) dw_done[wave] = s + 1;                  // every read of image and package has returned
                        RB_T(t5);
                        RB_ACC(4, t4, t5);
                    }
                }
            }
        }
#endif
    } else if constexpr (DS) {
    // ================= bands of small batches, two slices per round (DUAL SLICES above) =================
    constexpr int NRD = NCH / 2;
    constexpr int OFF_EG = OFF_PKG, OFF_DG = OFF_PKG + 2 * PK::EG_BYTES;
    static_assert(OFF_DG + 2 * PK::DG_BYTES == OFF_PKG + 2 * PK::BYTES, "the two packages' groups fill the two package buffers");
    // group `dgroup` of package p into the buffer of the round's slice `par`; piece k is copied by wave k % NW
    auto stage_group = [&](int p, int par, bool dgroup) {
        const char* g = WP + (size_t)p * PK::BYTES + (dgroup ? PK::EG_BYTES : 0) + lane * 16;
        char* l = lds + (dgroup ? OFF_DG + par * PK::DG_BYTES : OFF_EG + par * PK::EG_BYTES);
        const int pieces = dgroup ? PK::DG_PIECES : PK::EG_PIECES;
#pragma unroll
        for (int j = 0; j < (cmax(PK::DG_PIECES, PK::EG_PIECES) + NW - 1) / NW; ++j) {
            const int k = wave + NW * j;
            if (k < pieces)                                // wave-uniform
                __builtin_amdgcn_global_load_lds((GPTR)(g + k * 1024), (LPTR)(l + k * 1024), 16, 0, 0);
        }
    };
    const int e_base0 = OFF_EG, e_base1 = OFF_EG + PK::EG_BYTES;
    const int d_base = OFF_DG + dpar * PK::DG_BYTES - PK::EG_BYTES;    // where this wave's slice's package would start
    char* const xslot = lds + OFF_XCH + (wt * 2 + en) * XCH_SLOT;        // this (tile, subtile)'s hand-over slot
    stage_group(0, 0, false);
    stage_group(1, 1, false);
    for (int img = blockIdx.y; img < B; img += gridDim.y) {
        const bool has_next = img + (int)gridDim.y < B;
        load_x(img);
        if (img == (int)blockIdx.y) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // expand groups of packages 0 and 1 (later images: requested in the image before)
            __syncthreads();                                   // ... and the zeroed images
        }
        int p = 0;                                             // package of the round's even slice
#pragma unroll 1
        for (int b = 0; b < nblk; ++b) {
            const int blk = b;
#pragma unroll
            for (int n = 0; n < NT16; ++n) pacc[0][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bool dwa = wt < MTO + CH - 1 - blk;          // wave-uniform: this wave has depthwise / project work in this block
#pragma unroll 1
            for (int r = 0; r < NRD; ++r, p += 2) {
                RB_T(ts);
                // this round's depthwise / project groups: their buffers were last read before the previous round's last barrier
                stage_group(p, 0, true);
                stage_group(p + 1, 1, true);
                RB_T(t0);
                RB_ACC(5, ts, t0);
                expand(e_base0, 0, blk);
                expand(e_base1, ES_BYTES, blk);
                RB_T(t1);
                RB_ACC(0, t0, t1);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // both images written, both groups landed
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                RB_T(t2);
                RB_ACC(1, t1, t2);
                // the next round's expand groups (this round's have been read)
                const int pn = p + 2 < npk ? p + 2 : (has_next ? 0 : -1);
                if (pn >= 0) {
                    stage_group(pn, 0, false);
                    stage_group(pn + 1, 1, false);
                }
                dwproject(d_base, dpar * ES_BYTES, blk);
                RB_T(t3);
                RB_ACC(2, t2, t3);
                const bool block_end = r == NRD - 1;
                if (block_end && dwa && dpar == 1) {           // the odd slices' sums go to the wave that holds the tile's activation
#pragma unroll
                    for (int n = 0; n < NT16; ++n) *reinterpret_cast<f32x4*>(xslot + (n * 64 + lane) * 16) = pacc[0][n];
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                RB_T(t4);
                RB_ACC(3, t3, t4);
                __syncthreads();                               // images free, the next expand groups visible, hand-over written
                if (block_end) {
                    if (dwa && dpar == 0) {
#pragma unroll
                        for (int n = 0; n < NT16; ++n) {
                            const f32x4 o = *reinterpret_cast<const f32x4*>(xslot + (n * 64 + lane) * 16);
#pragma unroll
                            for (int i = 0; i < 4; ++i) pacc[0][n][i] += o[i];
                        }
                        block_epilogue(d_base, b == nblk - 1, img);            // reads b3 from the even slice's package
                        if (b + 1 < nblk) {                                    // this subtile's new activation for the wave it shares the tile with
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) {
                                if (en) *reinterpret_cast<V8*>(xslot + (ks * 64 + lane) * 16) = xf[0][1][ks];   // wave-uniform
                                else *reinterpret_cast<V8*>(xslot + (ks * 64 + lane) * 16) = xf[0][0][ks];
                            }
                        }
                    }
                    __syncthreads();                           // b3 read before the next round's groups arrive; hand-over of x written
                    if (b + 1 < nblk && dwa && dpar == 0) {
                        const char* xi = lds + OFF_XCH + (wt * 2 + (en ^ 1)) * XCH_SLOT;
                        if (en) {
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) xf[0][0][ks] = *reinterpret_cast<const V8*>(xi + (ks * 64 + lane) * 16);
                        } else {
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) xf[0][1][ks] = *reinterpret_cast<const V8*>(xi + (ks * 64 + lane) * 16);
                        }
                    }
                }
                RB_T(t5);
                RB_ACC(4, t4, t5);
            }
        }
    }
    } else {
    // A workgroup keeps its band and walks over images blockIdx.y, blockIdx.y + gridDim.y, ...
    for (int img = blockIdx.y; img < B; img += gridDim.y) {
        const bool has_next = img + (int)gridDim.y < B;
        load_x(img);
        if (img == (int)blockIdx.y) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // package 0 (later images: waited for at the previous image's end)
            __syncthreads();                                   // ... and the zeroed image
        }
        int p = 0;                                             // package of the current slice
#pragma unroll 1
        for (int b = 0; b < nblk; ++b) {
            const int blk = CH > 1 ? b : 0;
#pragma unroll
            for (int st = 0; st < NSUB; ++st)
#pragma unroll
                for (int n = 0; n < NT16; ++n) pacc[st][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int c = 0; c < NCH; ++c, ++p) {
                const int buf = c & 1;                         // NCH is even: the parity carries across blocks and images
                // next package into the other buffer (last read before the previous barrier)
                RB_T(ts);
#ifndef RB_ABL_NO_STAGE
                if (p + 1 < npk) stage(p + 1, buf ^ 1);
                else if (has_next) stage(0, buf ^ 1);
#endif
                RB_T(t0);
                RB_ACC(5, ts, t0);
#ifndef RB_ABL_NO_EXPAND
                expand(OFF_PKG + buf * PK::BYTES, 0, blk);
#endif
                RB_T(t1);
                RB_ACC(0, t0, t1);
                // the image of this slice is complete.  Not __syncthreads(): its fence would also wait for the LDS-DMA in flight
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef RB_ABL_NO_BARRIER1
                __builtin_amdgcn_s_barrier();
#endif
                asm volatile("" ::: "memory");
                RB_T(t2);
                RB_ACC(1, t1, t2);
#ifndef RB_ABL_NO_DW
                dwproject(OFF_PKG + buf * PK::BYTES, 0, blk);
#endif
                RB_T(t3);
                RB_ACC(2, t2, t3);
                const bool dwa = CH == 1 ? dw_active : (we < MTO + CH - 1 - blk && en < 2);      // wave-uniform
                if (c == NCH - 1 && dwa) block_epilogue(OFF_PKG + buf * PK::BYTES, b == nblk - 1, img);
                if constexpr (CH > 1) {
                    // the block's output for the next block's expand stage: this wave holds ITS subtile of the tile, the
                    // wave it shares the tile with the other one
                    if (c == NCH - 1 && b + 1 < nblk && dwa) {
                        char* xo = lds + OFF_XCH + ((we * 2 + en) * 64 + lane) * (KS * 16);
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            if (en) *reinterpret_cast<V8*>(xo + 16 * ks) = xf[0][1][ks];      // wave-uniform
                            else *reinterpret_cast<V8*>(xo + 16 * ks) = xf[0][0][ks];
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                RB_T(t4);
                RB_ACC(3, t3, t4);
#ifndef RB_ABL_NO_BARRIER2
                __syncthreads();                               // image free, next package visible
#endif
                if constexpr (CH > 1) {
                    if (c == NCH - 1 && b + 1 < nblk && dwa) {
                        const char* xi = lds + OFF_XCH + ((we * 2 + (en ^ 1)) * 64 + lane) * (KS * 16);
                        if (en) {
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) xf[0][0][ks] = *reinterpret_cast<const V8*>(xi + 16 * ks);
                        } else {
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) xf[0][1][ks] = *reinterpret_cast<const V8*>(xi + 16 * ks);
                        }
                    }
                }
                RB_T(t5);
                RB_ACC(4, t4, t5);
            }
        }
    }
    }
#ifdef RB_STAMPS
    RB_T(t_end);
    dbg[7] = t_end - t_begin;
    if (lane == 0)
        for (int i = 0; i < 8; ++i) fra_rb_cycles[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8 + i] = dbg[i];
#endif
}

// ---------------------------------------------------------------------------------------------
static int rb_resident_workgroups(const void* kernel, int threads, int lds_bytes) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, (size_t)lds_bytes) != hipSuccess || per_cu <= 0)
        per_cu = 1;
    return cus * per_cu;
}

static int rb_compute_units() {
    static std::atomic<int> cus_of[64];
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev < 64 && (cus = cus_of[dev].load(std::memory_order_relaxed)) > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (dev < 64) cus_of[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

template <typename T, int C, int CEXP, int H, int TR, int NW, int G = 1, int CH = 1, bool DUAL = false>
static int launch_rblock_cfg(const void* x, const void* wp, int nblk, void* out, int B, hipStream_t s) {
    constexpr int bands = H / TR;
    constexpr int lds = rblock_lds_bytes<C, CEXP, H, TR, NW, G, CH, DUAL>();
    FRA_REQUIRE(CH > 1 ? nblk == CH : (nblk == 1 || (TR == H && rb_main_tiles<H, TR, G>() > NW / 2)), FRA_ERR_ARG,
                "rblock: %d blocks per launch need whole-image workgroups with a tile per wave", nblk);
    static_assert(G == 1 || rb_flags<H, TR, NW, G>(), "stacked images run on the flag-synchronised loop");
    void (*kern)(const T*, const char*, T*, int, int, unsigned*, int);
    if constexpr (CH > 1) kern = rblock_kernel_w2<T, C, CEXP, H, TR, NW, G, CH, DUAL>;
    else kern = rblock_kernel<T, C, CEXP, H, TR, NW, G, CH>;
    static std::atomic<uint64_t> lds_set;      // per instantiation: devices whose LDS limit is raised
    static std::atomic<int> slots_of[64];      // resident workgroups per band, per device (0 = not asked yet)
    int dev = 0;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set, &dev)) return rc;
    int slots = dev < 64 ? slots_of[dev].load(std::memory_order_relaxed) : 0;
    if (slots == 0) {
        slots = rb_resident_workgroups(reinterpret_cast<const void*>(kern), NW * 64, lds) / bands;
        if (slots < 1) slots = 1;
        if (dev < 64) slots_of[dev].store(slots, std::memory_order_relaxed);
    }
    // one workgroup per band and resident slot, each walking over its images (groups of G images) in equal shares
    const int units = (B + G - 1) / G;
    const int per = (units + slots - 1) / slots;
    const int groups = (units + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3(bands, groups), dim3(NW * 64), lds, s, (const T*)x, (const char*)wp, (T*)out, B, nblk,
                       fault_word(), spin_limit(RB_SPIN_LIMIT));
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

#ifndef RB_STACK2_PER_CU
#define RB_STACK2_PER_CU 1   // pairs of stacked 7 x 7 images per compute unit up to which the two-image workgroups run
#endif
#ifndef RB_NW_WHOLE
#define RB_NW_WHOLE 8      // waves of a whole-image workgroup (7 tiles of two rows; the eighth wave only stages)
#endif
#ifndef RB_DUAL_ROUNDS
#define RB_DUAL_ROUNDS 1   // rounds of band workgroups over the chip up to which a batch runs the several-blocks-per-band launches in
                           // dual-slice mode.  Measured with 2 and 3: forward at 37 / 64 / 73 images 245 / 253 / 332 us against
                           // 220 / 225 / 251 us with whole-image workgroups from 37 on -- a second round doubles a launch's latency
#endif
// band workgroups (7 per image) that count as "fit the chip": one round, or RB_DUAL_ROUNDS in dual-slice mode
static int64_t rb_band_capacity(int cus) { return small_batch_mode() == 1 ? (int64_t)RB_DUAL_ROUNDS * cus : cus; }

// conv_5 (128 -> 256 -> 128 on 7 x 7), workgroup shape by batch size: 0 = one-row bands (7 workgroups per image, one
// block per launch) while those fit the chip in one go; 2 = two stacked images per workgroup (four tile waves) while
// a compute unit gets at most RB_STACK2_PER_CU such pairs; 4 = four stacked images (seven tile waves).  2 and 4 chain
// blocks in one launch.  (A whole image per workgroup with two waves per tile, one block per launch, was measured:
// 29 us for the two blocks at 37 to 256 images, what the two-image chain takes.)  FRA_RB_STACK=0|2|4 forces a shape
// (measurements).
static int rb_stack_choice(int B, int cus) {
    static const int forced = [] { const char* e = getenv("FRA_RB_STACK"); return e ? atoi(e) : -1; }();
    if (forced == 0 || forced == 2 || forced == 4) return forced;
    if ((int64_t)B * 7 <= rb_band_capacity(cus)) return 0;
    return B <= 2 * RB_STACK2_PER_CU * cus ? 2 : 4;
}

// FRA_RB_BAND_CHAIN=0: small batches run one block per launch (measurements)
static bool rb_band_chain() {
    static const bool on = [] { const char* e = getenv("FRA_RB_BAND_CHAIN"); return !e || atoi(e) != 0; }();
    return on;
}

// FRA_WBLOCK=1: whole-image workgroups of 14 x 14 maps run as four wide waves (kernels_wblock.hip) instead of eight
// waves with a tile each (this file); same bits either way, 12 % slower at batch 1024 (measurements, cross-checks)
static bool rb_wide_waves() {
    static const bool on = [] { const char* e = getenv("FRA_WBLOCK"); return e && atoi(e) != 0; }();
    return on;
}

// conv_4 (128 -> 256 -> 128 on 14 x 14): bands of 2 rows (7 workgroups per image) while those fit the chip in
// one go, bands of 7 rows (2 per image) while those do, whole images beyond
template <typename T>
static int launch_rblock_t(int c, int cexp, int h, const void* x, const void* wp, int nblk, void* out, int B, hipStream_t s) {
    const int cus = rb_compute_units();
    if (c == 128 && cexp == 256 && h == 14) {
        // two blocks per launch of two-row bands (each band computes its neighbours' nearest rows itself); not in place:
        // other workgroups read the input rows around their bands
        if (nblk == 2 && (int64_t)B * 7 <= rb_band_capacity(cus) && x != out && rb_band_chain())
            return small_batch_mode() == 1 ? launch_rblock_cfg<T, 128, 256, 14, 2, 8, 1, 2, true>(x, wp, 2, out, B, s)
                                           : launch_rblock_cfg<T, 128, 256, 14, 2, 8, 1, 2>(x, wp, 2, out, B, s);
        const bool wide = rb_wide_waves() && wblock_has(c, cexp, h);
        const int dt = std::is_same<T, f16>::value ? ACT_F16 : ACT_BF16;
        if (nblk > 1) return wide ? launch_wblock(dt, c, cexp, h, x, wp, nblk, out, B, s)
                                  : launch_rblock_cfg<T, 128, 256, 14, 14, RB_NW_WHOLE>(x, wp, nblk, out, B, s);
        if ((int64_t)B * 7 <= cus) return launch_rblock_cfg<T, 128, 256, 14, 2, 8>(x, wp, 1, out, B, s);
        if ((int64_t)B * 2 <= cus) return launch_rblock_cfg<T, 128, 256, 14, 7, 8>(x, wp, 1, out, B, s);
        return wide ? launch_wblock(dt, c, cexp, h, x, wp, 1, out, B, s)
                    : launch_rblock_cfg<T, 128, 256, 14, 14, RB_NW_WHOLE>(x, wp, 1, out, B, s);
    }
    if (c == 128 && cexp == 256 && h == 7) {
        int g = rb_stack_choice(B, cus);
        // conv_5's two blocks as one launch of one-row bands (each band computes the rows above and below it itself)
        if (g == 0 && nblk == 2 && x != out && rb_band_chain())
            return small_batch_mode() == 1 ? launch_rblock_cfg<T, 128, 256, 7, 1, 8, 1, 2, true>(x, wp, 2, out, B, s)
                                           : launch_rblock_cfg<T, 128, 256, 7, 1, 8, 1, 2>(x, wp, 2, out, B, s);
        if (nblk > 1 && g < 2) g = 2;                      // several blocks per launch otherwise: only the stacked shapes chain
        if (g == 0) return launch_rblock_cfg<T, 128, 256, 7, 1, 8>(x, wp, 1, out, B, s);
        if (g == 2) return launch_rblock_cfg<T, 128, 256, 7, 7, 5, 2>(x, wp, nblk, out, B, s);
        return launch_rblock_cfg<T, 128, 256, 7, 7, 8, 4>(x, wp, nblk, out, B, s);
    }
    FRA_REQUIRE(false, FRA_ERR_ARG, "rblock: no kernel for c=%d cexp=%d h=%d", c, cexp, h);
    return FRA_ERR_ARG;
}

int rblock_chain_blocks(int c, int cexp, int h, int B, int remaining) {
    if (remaining < 2) return 1;
    if (rblock_chain_pays(c, cexp, h, B)) return remaining;                 // whole images: the rest of the run in one launch
    if (c == 128 && cexp == 256 && (h == 14 || h == 7) && (int64_t)B * 7 <= rb_band_capacity(rb_compute_units()) && rb_band_chain()) return 2;
    return 1;
}

bool rblock_chain_pays(int c, int cexp, int h, int B) {
    // one launch of whole-image workgroups takes about as long for one image as for a chip full of them; it wins over
    // per-block launches as soon as the two-row bands of a batch no longer fit the chip in one go
    // (FRA_RB_CHAIN_MIN=<batch> moves the limit, for measurements)
    static const int min_b = [] { const char* e = getenv("FRA_RB_CHAIN_MIN"); return e ? atoi(e) : 0; }();
    if (c == 128 && cexp == 256 && h == 7) return rb_stack_choice(B, rb_compute_units()) >= 2;
    const bool shape = c == 128 && cexp == 256 && h == 14;
    if (min_b > 0) return shape && B >= min_b;
    return shape && (int64_t)B * 7 > rb_band_capacity(rb_compute_units());
}

int launch_rblock(int dt, int c, int cexp, int h, const void* x, const void* wp, int nblk, void* out, int B, hipStream_t s) {
    FRA_REQUIRE(x && wp && out, FRA_ERR_ARG, "rblock: null pointer");
    FRA_REQUIRE(B > 0 && B <= 65535, FRA_ERR_ARG, "rblock: batch %d not in [1,65535]", B);
    FRA_REQUIRE(nblk >= 1 && nblk <= 64, FRA_ERR_ARG, "rblock: %d blocks", nblk);
    FRA_REQUIRE(nblk > 1 || x != out, FRA_ERR_ARG, "rblock: a single block cannot run in place (bands read their neighbours' rows)");
    FRA_REQUIRE((((uintptr_t)wp) & 15) == 0, FRA_ERR_ARG, "rblock: packages must be 16-byte aligned");
    if (dt == ACT_F16) return launch_rblock_t<f16>(c, cexp, h, x, wp, nblk, out, B, s);
    if (dt == ACT_BF16) return launch_rblock_t<bf16>(c, cexp, h, x, wp, nblk, out, B, s);
    FRA_REQUIRE(false, FRA_ERR_ARG, "rblock: unsupported activation dtype %d", dt);
    return FRA_ERR_ARG;
}

// ---- host side: folded weights -> packages --------------------------------------------------------------------
size_t rblock_package_bytes(int c, int cexp) {
    if (cexp % 128 != 0) return 0;
    if (c == 128) return (size_t)(cexp / 64) * RbPkg<128>::BYTES;
    if (c == 64) return (size_t)(cexp / 64) * RbPkg<64>::BYTES;
    return 0;
}

static uint16_t rb_to16(int dt, float f) {
    if (dt == ACT_BF16) {
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
        return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    _Float16 hv = (_Float16)f;
    uint16_t b;
    memcpy(&b, &hv, 2);
    return b;
}

template <int C>
static void pack_rblock_c(int dt, int cexp, const float* w1, const float* b1, const float* s1, const float* w2,
                          const float* b2, const float* s2, const float* w3, const float* b3, unsigned char* out) {
    typedef RbPkg<C> PK;
    const int nch = cexp / 64;
    memset(out, 0, (size_t)nch * PK::BYTES);
    for (int c = 0; c < nch; ++c) {
        unsigned char* pk = out + (size_t)c * PK::BYTES;
        for (int el = 0; el < 64; ++el) {
            const int e = 64 * c + el, t = el >> 4, m = el & 15, pos = rb_pos(el);
            for (int par = 0; par < 2; ++par) {
                uint16_t* row = reinterpret_cast<uint16_t*>(pk + (size_t)(32 * t + 16 * par + m) * PK::PW1);
                for (int ks = 0; ks < C / 32; ++ks)
                    for (int gh = 0; gh < 2; ++gh)
                        for (int j = 0; j < 8; ++j)
                            row[(2 * ks + gh) * 8 + j] = rb_to16(dt, w1[(size_t)e * C + rb_in_channel(ks, 2 * gh + par, j)]);
            }
            reinterpret_cast<float*>(pk + PK::OFF_B1)[el] = b1[e];
            reinterpret_cast<float*>(pk + PK::OFF_S1)[el] = s1[e] - 1.0f;
            reinterpret_cast<float*>(pk + PK::OFF_B2)[pos] = b2[e];
            reinterpret_cast<float*>(pk + PK::OFF_S2)[pos] = s2[e] - 1.0f;
            for (int r = 0; r < 3; ++r) {                  // window row r: plane 0 = taps (r,0) | (r,1) << 16, plane 1 = tap (r,2)
                unsigned char* row = pk + PK::OFF_W2 + r * dw_wrow(64);
                reinterpret_cast<uint32_t*>(row)[pos] =
                    (uint32_t)rb_to16(dt, w2[(size_t)e * 9 + 3 * r]) | ((uint32_t)rb_to16(dt, w2[(size_t)e * 9 + 3 * r + 1]) << 16);
                reinterpret_cast<uint32_t*>(row + dw_wplane(64))[pos] = (uint32_t)rb_to16(dt, w2[(size_t)e * 9 + 3 * r + 2]);
            }
            const int P = (pos >> 5) & 1, hi = (pos >> 4) & 1, g = (pos >> 2) & 3, i = pos & 3;
            for (int co = 0; co < C; ++co) {
                uint16_t* row = reinterpret_cast<uint16_t*>(pk + PK::OFF_W3 + (size_t)(32 * (co >> 4) + 16 * (g & 1) + (co & 15)) * PK::PW3);
                row[(2 * P + (g >> 1)) * 8 + 4 * hi + i] = rb_to16(dt, w3[(size_t)co * cexp + e]);
            }
        }
        for (int co = 0; co < C; ++co) reinterpret_cast<float*>(pk + PK::OFF_B3)[co] = b3[co];
    }
}

int pack_rblock(int dt, int c, int cexp, const float* w1, const float* b1, const float* s1, const float* w2,
                const float* b2, const float* s2, const float* w3, const float* b3, void* out) {
    FRA_REQUIRE(w1 && b1 && s1 && w2 && b2 && s2 && w3 && b3 && out, FRA_ERR_ARG, "pack_rblock: null pointer");
    FRA_REQUIRE(dt == ACT_F16 || dt == ACT_BF16, FRA_ERR_ARG, "pack_rblock: dtype %d", dt);
    FRA_REQUIRE(rblock_package_bytes(c, cexp) > 0, FRA_ERR_ARG, "pack_rblock: unsupported widths c=%d cexp=%d", c, cexp);
    if (c == 128) pack_rblock_c<128>(dt, cexp, w1, b1, s1, w2, b2, s2, w3, b3, static_cast<unsigned char*>(out));
    else pack_rblock_c<64>(dt, cexp, w1, b1, s1, w2, b2, s2, w3, b3, static_cast<unsigned char*>(out));
    return FRA_OK;
}

}  // namespace fra
