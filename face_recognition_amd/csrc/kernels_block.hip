// Fused inverted-residual block for gfx950:  1x1 expand + PReLU -> 3x3 depthwise + PReLU ->
// 1x1 project (+ shortcut), one kernel, the expanded tensor never leaves the CU.
// (reference: Depth_Wise.forward, face_model.py:55-65)
//
// A workgroup owns a band of TR output rows of one image; every wave owns one 32-pixel tile of the
// band's input (expand side) and one of its output (depthwise / project side).  Per 64-channel
// slice of the expanded tensor the workgroup
//   1. runs the expand GEMM on the matrix cores for every input pixel of the band (+ halo rows),
//      with the input tile's MFMA fragments held in registers for the whole kernel,
//   2. writes PReLU(expand) as 16-bit values into an LDS image [row][x+1][64 ch] whose border
//      columns and out-of-image rows stay zero (that IS the depthwise conv's zero padding),
//   3. lets every lane build, for ITS output pixel, the 3x3 depthwise result of 8 channels from
//      nine 16-byte LDS reads, and feeds those 8 values straight to the project MFMA as its B
//      fragment -- the depthwise output never exists in memory at all.
// Project accumulators live in registers across slices.  The epilogue turns them into rows of
// pixels through LDS so that the shortcut is read and the result written with contiguous 8-byte
// pieces per lane.  The weight-matrix slices of the next 64 channels are fetched into registers
// while the current slice computes and stored into the other of two LDS buffers.
//
// LDS layout.  Every tile that is read as MFMA fragments or by the depthwise stage has rows of
// 16-byte units with one unit of padding (a pixel slot of the image is 128 + 16 bytes, a weight row
// CIN*2 + 16 or 128 + 16 bytes): 16 lanes that read the same unit of 16 consecutive rows then hit
// 16 different 16-byte columns of the 256-byte bank window, and every unit / k-step / tap offset
// is a compile-time constant that goes into the instruction's offset field.
//
// The LDS reads of both stages are issued by hand ahead of their use, with counted waits: LDS
// returns data in order, so "at most n reads outstanding" (s_waitcnt lgkmcnt(n)) says exactly
// which read has arrived.  A wait that allows fewer outstanding reads than necessary is always
// safe; one that allows more is a bug -- the counts below are derived from the issue order stated
// next to them.
//
// Channel order inside the expanded tensor: a lane of half h of the expand MFMA ends up with
// channels 8q+4h+i (q<4, i<4) of each 32-channel tile, and the project MFMA wants, for k-step s,
// element j of half h to be one fixed channel.  Taking channel 16s + 8(j>>2) + 4h + (j&3) makes
// the two coincide, so a lane's 8 values per 16-channel group travel as ONE 16-byte unit from the
// expand epilogue through LDS and the depthwise stage into the project MFMA.  Everything indexed
// by expanded channel on the depthwise/project side (bias, slope, 3x3 weights, project weight
// columns) is stored pre-permuted on the host: natural channel 16g + 8a + 4h + i sits at
// position 16g + 8h + 4a + i.  The two slope arrays hold (PReLU slope - 1): prelu(v) = v + (slope-1) min(v,0).
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

// ---- issue order of the depthwise stage (one slice = 4 k-steps s4 x 9 taps = 36 steps) ------------
//   preamble : late header of s4 = 0 (project fragments a3[NT3], slopes s2a s2b), bias pair of
//              s4 = 0, taps of steps 0 .. PD-1 (two reads each: image unit, 3x3 weights)
//   step i   : taps of step i + PD; if i % 9 == 4 the bias pair of the next s4;  WAIT for step i;
//              8 FMAs; if i % 9 == 8: PReLU, project MFMAs, then the late header of the next s4
// dw_ahead(i) = reads issued after the taps of step i and before its wait.
constexpr int DW_STEPS = 36, DW_PD = 3, DW_RING = DW_PD + 1;
constexpr int dw_ahead(int step, int nt3) {
    int n = 2 * (cmin(step + DW_PD, DW_STEPS - 1) - step);
    for (int j = 4; j < DW_STEPS - 9; j += 9)          // bias pairs go out in steps 4, 13, 22
        if (j <= step && j >= step - DW_PD) n += 2;
    for (int e = 8; e < DW_STEPS - 9; e += 9)          // late headers after steps 8, 17, 26
        if (e < step && e >= step - DW_PD) n += nt3 + 2;
    return cmin(n, 15);
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

// ---- depthwise + project stage of one slice (see the issue order above dw_ahead) ------------------
template <typename T, int CEXP, int NT3, int PW3> struct DwStage {
    typedef typename vec8<T>::type V8;
    V8 er[DW_RING], wr[DW_RING];   // ring: image unit and 3x3 weights of a tap
    V8 a3[NT3];                    // project fragments of the current k-step
    f32x4 s2a, s2b, b2[2][2];
    float d[8];
    int tapb[9];   // LDS address of unit h of this lane's 9 taps; + 32 s4
    int w2l;       // 3x3 weights of the slice for half h; + tap CEXP*2 + 32 s4
    int pb;        // depthwise bias quads of the slice for half h; + 64 s4 (+16); slopes CEXP*4 further
    int w3l;       // this lane's W3 row (+ 16 h); + 32 n3 PW3 + 32 s4

    template <int ST> __device__ __forceinline__ void issue_tap() {
        constexpr int s4 = ST / 9, k = ST % 9;
        LDS_READ16(er[ST % DW_RING], tapb[k], 32 * s4);
        LDS_READ16(wr[ST % DW_RING], w2l, k * CEXP * 2 + 32 * s4);
    }
    template <int S4, int N3 = 0> __device__ __forceinline__ void issue_late() {
        if constexpr (N3 < NT3) {
            LDS_READ16(a3[N3], w3l, N3 * 32 * PW3 + 32 * S4);
            issue_late<S4, N3 + 1>();
        } else {
            LDS_READ16(s2a, pb, CEXP * 4 + 64 * S4);
            LDS_READ16(s2b, pb, CEXP * 4 + 64 * S4 + 16);
        }
    }
    template <int S4> __device__ __forceinline__ void issue_bias() {
        LDS_READ16(b2[S4 & 1][0], pb, 64 * S4);
        LDS_READ16(b2[S4 & 1][1], pb, 64 * S4 + 16);
    }
    template <int ST = 0> __device__ __forceinline__ void preamble() {
        if constexpr (ST == 0) { issue_late<0>(); issue_bias<0>(); }
        if constexpr (ST < DW_PD) { issue_tap<ST>(); preamble<ST + 1>(); }
    }
    template <int N3 = 0> __device__ __forceinline__ void pin_late() {
        if constexpr (N3 < NT3) { asm volatile("" : "+v"(a3[N3])); pin_late<N3 + 1>(); }
        else asm volatile("" : "+v"(s2a), "+v"(s2b));
    }
    template <int ST = 0> __device__ __forceinline__ void run(f32x16 (&pacc)[NT3]) {
        if constexpr (ST < DW_STEPS) {
            constexpr int s4 = ST / 9, k = ST % 9, slot = ST % DW_RING;
            if constexpr (ST + DW_PD < DW_STEPS) issue_tap<ST + DW_PD>();
            if constexpr (k == 4 && s4 + 1 < 4) issue_bias<s4 + 1>();
            if constexpr (k == 0) {
                // the bias pair went out before this step's taps, so it has arrived with them
                asm volatile("s_waitcnt lgkmcnt(%4)"
                             : "+v"(er[slot]), "+v"(wr[slot]), "+v"(b2[s4 & 1][0]), "+v"(b2[s4 & 1][1])
                             : "n"(dw_ahead(ST, NT3)));
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    d[j] = fmaf((float)er[slot][j], (float)wr[slot][j], b2[s4 & 1][j >> 2][j & 3]);
            } else {
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(er[slot]), "+v"(wr[slot]) : "n"(dw_ahead(ST, NT3)));
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = fmaf((float)er[slot][j], (float)wr[slot][j], d[j]);
            }
            if constexpr (k == 8) {
                // the late header went out before this k-step's tap PD, so it arrived before tap 8 did
                pin_late();
                V8 bf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf[j] = (T)fmaf(fminf(d[j], 0.0f), s2a[j], d[j]);               // PReLU: s2 = slope - 1
                    bf[4 + j] = (T)fmaf(fminf(d[4 + j], 0.0f), s2b[j], d[4 + j]);
                }
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3) pacc[n3] = mfma32(a3[n3], bf, pacc[n3]);
                if constexpr (s4 + 1 < 4) {
                    __builtin_amdgcn_sched_barrier(0);     // the header registers are free once the MFMAs are issued
                    issue_late<s4 + 1>();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            run<ST + 1>(pacc);
        }
    }
};

template <typename T, int CIN, int CEXP, int COUT, int H, int S, int TR, bool RESID, int NW, int MINW, int MODE>
__global__ __launch_bounds__(NW * 64, MINW)
void block_kernel(const T* __restrict__ X, const T* __restrict__ W1, const float* __restrict__ b1p,
                  const float* __restrict__ s1p, const T* __restrict__ W2p, const float* __restrict__ b2p,
                  const float* __restrict__ s2p, const T* __restrict__ W3p, const float* __restrict__ b3,
                  T* __restrict__ O, int B) {
    typedef typename vec8<T>::type V8;
    typedef typename vec4<T>::type V4;
    constexpr int NT = NW * 64;
    constexpr int W = H, Ho = H / S, Wo = Ho;
    constexpr bool WHOLE = (TR == Ho);                     // the band is the whole image: no halo rows to compute
    constexpr int RIN = (S == 1) ? TR + 2 : 2 * TR + 1;    // LDS rows (with halo)
    constexpr int RCOMP = WHOLE ? H : RIN;                 // input rows the expand GEMM runs on
    constexpr int ROFF = WHOLE ? 1 : 0;                    // LDS row of the first computed input row
    constexpr int PIN = RCOMP * W, POUT = TR * Wo;
    // Output tiles are row aligned: a tile holds RPT whole output rows, each padded to POW (a power of two)
    // lanes, so that the pixel slots a 16-lane group reads are consecutive -- a tile of 32 consecutive pixels
    // would skip the two border slots at every row end and put two lanes on one bank.
    constexpr int POW = Wo <= 8 ? 8 : (Wo <= 16 ? 16 : 32), RPT = 32 / POW;
    constexpr int MTI = (PIN + 31) / 32, MTO = (TR + RPT - 1) / RPT;
    constexpr int KS1 = CIN / 16, NT3 = COUT / 32, NCH = CEXP / 64;
    constexpr int PITCH = W + 2;                           // pixel slots per LDS row (zero border each side)
    constexpr int SLOT = 144;                              // bytes per pixel slot: 64 channels + one unit of padding
    constexpr int PW1 = CIN * 2 + 16, PW3 = 144;           // padded row pitch of the W1 / W3 slices
    constexpr int ES_BYTES = RIN * PITCH * SLOT;
    constexpr int W1_BYTES = 64 * PW1, W3_BYTES = COUT * PW3;
    constexpr int NWB = MODE >= 1 ? 2 : 1;                 // LDS copies of each weight slice
    constexpr int OFF_W1 = ES_BYTES, OFF_W3 = OFF_W1 + NWB * W1_BYTES, OFF_PAR = OFF_W3 + NWB * W3_BYTES;
    constexpr int OFF_B1 = OFF_PAR, OFF_S1 = OFF_B1 + CEXP * 4, OFF_B2 = OFF_S1 + CEXP * 4, OFF_S2 = OFF_B2 + CEXP * 4,
                  OFF_W2 = OFF_S2 + CEXP * 4;
    constexpr int MAIN_BYTES = OFF_W2 + 9 * CEXP * 2;
    constexpr int P32 = COUT * 4 + 16;                     // epilogue staging: f32 pixel rows, padded
    constexpr int EPI_BYTES = POUT * P32;                  // overlays the image and weight buffers
    constexpr int LDS_BYTES = cmax(MAIN_BYTES, EPI_BYTES);
    constexpr int CH1 = CIN / 8;
    constexpr int W1_V = 64 * CH1, W3_V = COUT * 8;        // 16-byte vectors per slice
    constexpr int W1_PT = (W1_V + NT - 1) / NT, W3_PT = (W3_V + NT - 1) / NT;
    static_assert(CEXP % 64 == 0 && CIN % 16 == 0 && COUT % 32 == 0 && Ho % TR == 0, "shape");
    constexpr int MTIW = (MTI + NW - 1) / NW;              // input tiles per wave (stride-2 bands have more input than output)
    static_assert(MTO <= NW, "one output tile per wave");
    static_assert(!RESID || (S == 1 && CIN == COUT), "shortcut needs matching shapes");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(NT % (COUT / 4) == 0, "epilogue: a thread keeps one channel group");
    static_assert(EPI_BYTES <= OFF_PAR, "the epilogue staging must not reach the per-channel parameters");
    static_assert(NCH >= 2, "the weight pipeline below assumes at least two slices");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int band = blockIdx.x;
    const int y0 = band * TR;                              // first output row of the band
    const int row0 = WHOLE ? 0 : ((S == 1) ? y0 - 1 : 2 * y0 - 1);   // image row of the first computed input row

    // ---- weight-slice staging: global -> registers (issued a slice ahead) -> padded LDS rows ----------
    V8 wreg1[W1_PT], wreg3[W3_PT];
    auto fetch_w = [&](int c) {
#pragma unroll
        for (int i = 0; i < W1_PT; ++i) {
            const int v = tid + i * NT;
            if (v < W1_V) wreg1[i] = *reinterpret_cast<const V8*>(W1 + (size_t)c * 64 * CIN + (size_t)v * 8);
        }
#pragma unroll
        for (int i = 0; i < W3_PT; ++i) {
            const int v = tid + i * NT;
            if (v < W3_V) wreg3[i] = *reinterpret_cast<const V8*>(W3p + (size_t)(v >> 3) * CEXP + c * 64 + (v & 7) * 8);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W1_PT; ++i) {
            const int v = tid + i * NT;
            const int row = v / CH1, ch = v - row * CH1;
            if (v < W1_V) *reinterpret_cast<V8*>(lds + OFF_W1 + buf * W1_BYTES + row * PW1 + ch * 16) = wreg1[i];
        }
#pragma unroll
        for (int i = 0; i < W3_PT; ++i) {
            const int v = tid + i * NT;
            if (v < W3_V) *reinterpret_cast<V8*>(lds + OFF_W3 + buf * W3_BYTES + (v >> 3) * PW3 + (v & 7) * 16) = wreg3[i];
        }
    };
    fetch_w(0);

    // per-channel parameters of all slices
    for (int i = tid; i < CEXP / 4; i += NT) {
        reinterpret_cast<f32x4*>(lds + OFF_B1)[i] = reinterpret_cast<const f32x4*>(b1p)[i];
        reinterpret_cast<f32x4*>(lds + OFF_S1)[i] = reinterpret_cast<const f32x4*>(s1p)[i];
        reinterpret_cast<f32x4*>(lds + OFF_B2)[i] = reinterpret_cast<const f32x4*>(b2p)[i];
        reinterpret_cast<f32x4*>(lds + OFF_S2)[i] = reinterpret_cast<const f32x4*>(s2p)[i];
    }
    for (int i = tid; i < 9 * CEXP / 8; i += NT)
        reinterpret_cast<V8*>(lds + OFF_W2)[i] = reinterpret_cast<const V8*>(W2p)[i];

    // ---- input fragments (B operand of the expand GEMM), resident while an image's slices are expanded -------
    V8 xf[MTIW][KS1];
    int in_lds[MTIW];     // LDS byte address of this lane's input pixel slot (+ 16 h), -1 = nothing to write
    int x_off[MTIW];      // element offset of this lane's input pixel inside an image (+ 8 h)
#pragma unroll
    for (int t = 0; t < MTIW; ++t) {
        const int p = 32 * (wave + NW * t) + r;
        const int iy = p / W, ix = p - iy * W;
        const int gy = row0 + iy;
        const bool ok = (p < PIN) && (gy >= 0) && (gy < H);
        in_lds[t] = ok ? ((iy + ROFF) * PITCH + ix + 1) * SLOT + 16 * h : -1;
        x_off[t] = ((ok ? gy : 0) * W + (ok ? ix : 0)) * CIN + 8 * h;
    }
    auto load_x = [&](int img) {
        const T* Ximg = X + (size_t)img * H * W * CIN;
#pragma unroll
        for (int t = 0; t < MTIW; ++t)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) xf[t][ks] = *reinterpret_cast<const V8*>(Ximg + x_off[t] + 16 * ks);
    };
    load_x(blockIdx.y);

    // ---- output pixel of this lane: LDS byte address of unit h of each of its 9 taps ----------
    int tapb[9];
    const int oy = wave * RPT + r / POW, ox = r % POW;     // this lane's output pixel inside the band
    const bool out_ok = (ox < Wo) && (oy < TR);
    const int po = oy * Wo + ox;
    {
        const int cy = out_ok ? oy : 0, cx = out_ok ? ox : 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) tapb[ky * 3 + kx] = ((cy * S + ky) * PITCH + cx * S + kx) * SLOT + 16 * h;
    }

    // project accumulators: row acc_row(g,h) of tile n3 <-> channel 32 n3 + 8q + 4h + i; the bias joins in the epilogue
    f32x16 pacc[NT3];

    // ================= expand of slice c: E[64 ch][pixels] = PReLU(W1 . X + b1) -> LDS image =================
    typedef ExpStage<T, KS1, CEXP, PW1> Ex;
    auto expand_tile = [&](Ex& ex, int t, auto waitc, int nt, int eoff) {
        // one pixel tile against the fragments of channel tile nt that `ex` holds
        f32x16 acc;
        ex.template chain<decltype(waitc)::value>(acc, xf[t]);
        if (in_lds[t] >= 0) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                V8 o;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int q = 2 * s2 + a;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = acc[4 * q + i];
                        o[4 * a + i] = (T)fmaf(fminf(v, 0.0f), ex.sl[q][i], v);   // PReLU: sl = slope - 1
                    }
                }
                // 16-byte unit (2 nt + s2) * 2 + h of the slot
                *reinterpret_cast<V8*>(lds + eoff + in_lds[t] + (2 * nt + s2) * 32) = o;
            }
        }
    };
    auto expand = [&](int c, int wb, int eb) {
        if (wave >= MTI) return;                           // wave-uniform; such a wave has no later tile either
        const int eoff = eb * ES_BYTES;
        Ex ex;   // lives only inside this call, so its registers are free for the other stage
        ex.w1l = OFF_W1 + wb * W1_BYTES + r * PW1 + 16 * h;
        ex.pl = OFF_B1 + 256 * c + 32 * h;
        ex.template issue<0>();
        expand_tile(ex, 0, std::true_type{}, 0, eoff);
#pragma unroll
        for (int t = 1; t < MTIW; ++t)
            if (wave + NW * t < MTI) expand_tile(ex, t, std::false_type{}, 0, eoff);
        ex.template issue<1>();
        expand_tile(ex, 0, std::true_type{}, 1, eoff);
#pragma unroll
        for (int t = 1; t < MTIW; ++t)
            if (wave + NW * t < MTI) expand_tile(ex, t, std::false_type{}, 1, eoff);
    };

    // ================= depthwise 3x3 of slice c from the LDS image -> project MFMA =================================
    auto dwproject = [&](int c, int wb, int eb) {
        if (wave >= MTO) return;                           // wave-uniform
        DwStage<T, CEXP, NT3, PW3> dw;
#pragma unroll
        for (int k = 0; k < 9; ++k) dw.tapb[k] = tapb[k] + eb * ES_BYTES;
        dw.w2l = OFF_W2 + 128 * c + 16 * h;
        dw.pb = OFF_B2 + 256 * c + 32 * h;
        dw.w3l = OFF_W3 + wb * W3_BYTES + r * PW3 + 16 * h;
        dw.preamble();
        dw.run(pacc);
    };

    // A workgroup keeps its band and walks over images blockIdx.y, blockIdx.y + gridDim.y, ...: the per-channel
    // parameters are staged once, and the next image's input fragments and first weight slice are
    // requested while the current image's last slice and epilogue still run.
    static_assert(MODE == 0 || MODE == 1, "modes: 0 = one copy of the weight slices, 1 = two");
    for (int img = blockIdx.y; img < B; img += gridDim.y) {
        const bool has_next = img + (int)gridDim.y < B;
        // zero the activation image: borders and out-of-image rows are never written by the expand stage
        // (the previous image's epilogue used this memory as staging)
        for (int i = tid; i < ES_BYTES / 16; i += NT) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        store_w(0);
        fetch_w(1);
        __syncthreads();
#pragma unroll
        for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
            for (int g = 0; g < 16; ++g) pacc[n3][g] = 0.0f;
        if (MODE == 1) {
            // expand | barrier | depthwise+project, next slice's weights into the other buffer | barrier
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                expand(c, c & 1, 0);
                if (c == NCH - 1 && has_next) load_x(img + gridDim.y);   // the fragments are dead after the last expand
                __syncthreads();                       // the image of slice c is complete
                dwproject(c, c & 1, 0);
                if (c + 1 < NCH) {
                    store_w((c + 1) & 1);              // that buffer was last read before the previous barrier
                    // one load site for both cases (the next slice, or the next image's first slice, which is
                    // stored after the epilogue), so the loads stay here, a whole slice ahead of their use
                    if (c + 2 < NCH || has_next) fetch_w(c + 2 < NCH ? c + 2 : 0);
                }
                __syncthreads();                       // image and old weights free, new weights visible
            }
        } else {
            // one copy of the weight slices (less LDS, so that two workgroups share a CU and fill each
            // other's barrier waits): expand | barrier | depthwise+project | barrier | restage | barrier
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                expand(c, 0, 0);
                if (c == NCH - 1 && has_next) load_x(img + gridDim.y);
                __syncthreads();
                dwproject(c, 0, 0);
                __syncthreads();
                if (c + 1 < NCH) {
                    store_w(0);
                    if (c + 2 < NCH || has_next) fetch_w(c + 2 < NCH ? c + 2 : 0);
                    __syncthreads();
                }
            }
        }

        // ---- epilogue: accumulators -> f32 pixel rows in LDS -> (+ bias, + shortcut) -> NHWC, 8 bytes per lane -------
        if (wave < MTO && out_ok) {
            char* dst = lds + po * P32 + 16 * h;
    #pragma unroll
            for (int n3 = 0; n3 < NT3; ++n3)
    #pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(dst + (32 * n3 + 8 * q) * 4) =
                        f32x4{pacc[n3][4 * q], pacc[n3][4 * q + 1], pacc[n3][4 * q + 2], pacc[n3][4 * q + 3]};
        }
        __syncthreads();
        {
            constexpr int G = COUT / 4, UNITS = POUT * G;      // a unit = 4 channels of one pixel
            const int g = tid % G;
            const f32x4 bias = *reinterpret_cast<const f32x4*>(b3 + 4 * g);
            const size_t pix0 = (size_t)img * Ho * Wo + (size_t)y0 * Wo;
            const T* Xo = X + pix0 * CIN;
            T* Oo = O + pix0 * COUT;
    #pragma unroll 4
            for (int u = tid; u < UNITS; u += NT) {
                const int p = u / G;
                f32x4 v = *reinterpret_cast<const f32x4*>(lds + p * P32 + g * 16);
                if (RESID) {
                    const V4 rr = *reinterpret_cast<const V4*>(Xo + (size_t)p * CIN + 4 * g);
    #pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)rr[i];
                }
                V4 o;
    #pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (T)(v[i] + bias[i]);
                *reinterpret_cast<V4*>(Oo + (size_t)p * COUT + 4 * g) = o;
            }
        }
        if (has_next) __syncthreads();   // staging reads are done before the region is zeroed and restaged
    }
}

// ---------------------------------------------------------------------------------------------
template <int CIN, int CEXP, int COUT, int H, int S, int TR, int MODE>
constexpr int block_lds_bytes() {
    constexpr int RIN = (S == 1) ? TR + 2 : 2 * TR + 1;
    constexpr int main_bytes = RIN * (H + 2) * 144 +
                               (MODE >= 1 ? 2 : 1) * (64 * (CIN * 2 + 16) + COUT * 144) + 4 * CEXP * 4 + 9 * CEXP * 2;
    constexpr int epi_bytes = TR * (H / S) * (COUT * 4 + 16);
    return cmax(main_bytes, epi_bytes);
}

// workgroups of `kernel` that fit on the current device at once (occupancy x compute units)
static int resident_workgroups(const void* kernel, int threads, int lds_bytes) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, (size_t)lds_bytes) != hipSuccess || per_cu <= 0)
        per_cu = 1;
    return cus * per_cu;
}

template <typename T, int CIN, int CEXP, int COUT, int H, int S, int TR, bool RESID, int NW, int MINW, int MODE, bool PERSIST>
static int launch_block_cfg(const BlockParams& p, const void* x, void* out, int B, hipStream_t s) {
    FRA_REQUIRE(p.cout == COUT, FRA_ERR_ARG, "block: cout %d, fused kernel has %d", p.cout, COUT);
    // PERSIST: one workgroup per band and resident slot, each walking over the images blockIdx.y,
    // blockIdx.y + gridDim.y, ... in equal shares (the smallest group count that needs only
    // ceil(B / slots) images per workgroup).  Otherwise one workgroup per band and image, which balances
    // better when the bands do not divide the resident slots evenly.
    const int bands = (H / S) / TR;
    constexpr int lds = block_lds_bytes<CIN, CEXP, COUT, H, S, TR, MODE>();
    auto kern = block_kernel<T, CIN, CEXP, COUT, H, S, TR, RESID, NW, MINW, MODE>;
    static bool attr_done = false;   // per instantiation
    static int slots = 1;            // resident workgroups per band (of the device first used; one process drives one GPU)
    if (!attr_done) {
        FRA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        slots = resident_workgroups(reinterpret_cast<const void*>(kern), NW * 64, lds) / bands;
        if (slots < 1) slots = 1;
        attr_done = true;
    }
    int groups = B;
    if (PERSIST) {
        const int per = (B + slots - 1) / slots;
        groups = (B + per - 1) / per;
    }
    const dim3 grid(bands, groups), block(NW * 64);
    hipLaunchKernelGGL(kern, grid, block, lds, s, (const T*)x, (const T*)p.w1, p.b1p, p.s1p, (const T*)p.w2p, p.b2p,
                       p.s2p, (const T*)p.w3p, p.b3, (T*)out, B);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

template <typename T>
static int launch_block_t(const BlockParams& p, const void* x, void* out, int B, hipStream_t s) {
    const int key = p.cin * 1000000 + p.cexp * 1000 + p.h * 10 + p.stride;
    switch (key) {
        case 64 * 1000000 + 128 * 1000 + 56 * 10 + 2:   // conv_23
            return launch_block_cfg<T, 64, 128, 64, 56, 2, 4, false, 8, 2, 1, true>(p, x, out, B, s);
        case 64 * 1000000 + 128 * 1000 + 28 * 10 + 1:   // conv_3.*
            return launch_block_cfg<T, 64, 128, 64, 28, 1, 7, true, 8, 2, 1, true>(p, x, out, B, s);
        case 64 * 1000000 + 256 * 1000 + 28 * 10 + 2:   // conv_34
            return launch_block_cfg<T, 64, 256, 128, 28, 2, 7, false, 8, 2, 1, true>(p, x, out, B, s);
        case 128 * 1000000 + 256 * 1000 + 14 * 10 + 1:  // conv_4.*
            return launch_block_cfg<T, 128, 256, 128, 14, 1, 14, true, 8, 2, 1, true>(p, x, out, B, s);
        case 128 * 1000000 + 512 * 1000 + 14 * 10 + 2:  // conv_45
            return launch_block_cfg<T, 128, 512, 128, 14, 2, 7, false, 8, 2, 1, true>(p, x, out, B, s);
        case 128 * 1000000 + 256 * 1000 + 7 * 10 + 1:   // conv_5.*
            return launch_block_cfg<T, 128, 256, 128, 7, 1, 7, true, 8, 2, 1, true>(p, x, out, B, s);
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
