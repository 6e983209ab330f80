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
//   3. runs the 3x3 depthwise conv on the matrix cores as well: per 16-channel chunk and 16-pixel
//      subtile three sparse 16x16x64 MFMAs whose A operand carries the weights of a window row on its
//      diagonal and whose B operand is two 16-byte units of the LDS image per lane.  Bias enters as the C
//      operand; after PReLU and rounding a lane's four values are its share of the project MFMA's
//      B operand -- the depthwise output never exists in memory at all.
// Project accumulators live in registers across slices.  The epilogue turns them into rows of
// pixels through LDS so that the shortcut is read and the result written with contiguous 8-byte
// pieces per lane.  The weight-matrix slices of the next 64 channels arrive by LDS-DMA in the other of
// two LDS buffers while the current slice computes (no staging registers, no store pass).
//
// LDS layout.  Every tile that is read as MFMA fragments or by the depthwise stage has rows of
// 16-byte units with one unit of padding (a pixel slot of the image is 128 + 16 bytes, an expand
// weight row CIN*2 + 16, a project weight row 64 + 16 bytes): 16 lanes that read the same unit of 16
// consecutive rows then hit 16 different 16-byte columns of the 256-byte bank window, and every
// unit / k-step / tap offset is a compile-time constant that goes into the instruction's offset
// field.
//
// The LDS reads of both stages are issued by hand ahead of their use, with counted waits: LDS
// returns data in order, so "at most n reads outstanding" (s_waitcnt lgkmcnt(n)) says exactly
// which read has arrived.  A wait that allows fewer outstanding reads than necessary is always
// safe; one that allows more is a bug -- the counts below are derived from the issue order stated
// next to them.
//
// Channel order inside the expanded tensor: a lane of half h of the expand MFMA ends up with
// channels 8q+4h+i (q<4, i<4) of each 32-channel tile; storing natural channel 16g + 8a + 4h + i at
// position 16g + 8h + 4a + i lets it write them as 16-byte units.  Everything indexed by expanded
// channel on the depthwise/project side (bias, slope, 3x3 weights) is stored in that position order
// on the host; the project weights are stored as the LDS image of each slice (w3_image_index()).  The two slope arrays hold (PReLU slope - 1):
// prelu(v) = v + (slope-1) min(v,0).

#include <stdlib.h>

#include "block_stages.h"


#ifndef EPI_CHUNK
#define EPI_CHUNK 13     // epilogue: shortcut values are requested this many units ahead (13 = all of a 14-row conv_3 band's: -3 % against 4)
#endif

namespace fra {

template <int CIN, int CEXP, int COUT, int H, int S, int TR, int MODE, int KSPLIT> constexpr int block_lds_bytes();

template <typename T, int CIN, int CEXP, int COUT, int H, int S, int TR, bool RESID, int NW, int MINW, int MODE, int KSPLIT>
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
    constexpr int KS1 = CIN / 16, NT16 = COUT / 16, NCH = CEXP / 64;
    // pixel slots per LDS row (zero border each side).  The depthwise reads of a 16-lane group stay on distinct
    // banks when vertically neighbouring taps are a multiple of 16 slots apart (stride 1) or an odd number (stride 2)
    constexpr int PITCH = block_pitch<H, S>();
    constexpr int SLOT = 144;                              // bytes per pixel slot: 64 channels + one unit of padding
    // padded row pitch of the W1 slice; the W3 slice is stored as rows of 64 + 16 bytes: row 32 n + 16 (g & 1) + m
    // holds, for output channel 16 n + m, the units [pair][g >> 1] of k-group g (w3_image_index() in common.h)
    constexpr int PW1 = CIN * 2 + 16, PW3 = 80;
    constexpr int ES_BYTES = RIN * PITCH * SLOT;
    constexpr int W1_BYTES = 64 * PW1, W3_BYTES = 2 * COUT * PW3, WB_BYTES = W1_BYTES + W3_BYTES;
    constexpr int P32 = COUT * 4 + 16;                     // epilogue staging: f32 pixel rows, padded
    constexpr int EPI1 = POUT * P32;                       // one set of staging rows
    constexpr int EPI_BYTES = KSPLIT * EPI1;               // overlays the image and weight buffer 1
    // LDS map: image | weight buffer 1 (odd slices) | per-channel parameters | 3x3 weights | weight buffer 0 (even
    // slices).  The epilogue's staging rows overlay the image and buffer 1 only: buffer 0 can receive the next
    // image's first slice while this image's last slice and epilogue still run.
    constexpr int OFF_WB1 = ES_BYTES;
    constexpr int OFF_PAR = (cmax(OFF_WB1 + WB_BYTES, EPI_BYTES) + 15) / 16 * 16;
    constexpr int OFF_B1 = OFF_PAR, OFF_S1 = OFF_B1 + CEXP * 4, OFF_B2 = OFF_S1 + CEXP * 4, OFF_S2 = OFF_B2 + CEXP * 4,
                  OFF_W2 = OFF_S2 + CEXP * 4;
    constexpr int OFF_WB0 = OFF_W2 + dw_wimage_bytes(CEXP);   // 3x3 weight image (common.h)
    constexpr int LDS_BYTES = OFF_WB0 + WB_BYTES;
    constexpr int CH1 = CIN / 8;
    // a weight slice arrives by LDS-DMA in pieces of 64 lanes x 16 bytes = 1 KB of the padded LDS image
    constexpr int NI1 = W1_BYTES / 1024, NI3 = W3_BYTES / 1024;
    static_assert(W1_BYTES % 1024 == 0 && W3_BYTES % 1024 == 0, "weight images are whole DMA pieces");
    static_assert(MODE == 1, "two copies of the weight slices");
    static_assert(CEXP % 64 == 0 && CIN % 16 == 0 && COUT % 32 == 0 && Ho % TR == 0, "shape");
    constexpr int MTIW = (MTI + NW - 1) / NW;              // input tiles per wave (stride-2 bands have more input than output)
    // Expand side: wave -> (input tile we, channel tile en of the slice).  With at most four input tiles (ESPLIT) two
    // waves share a tile and expand one 32-channel tile of the slice each; otherwise a wave does both.
    constexpr bool ESPLIT = MTI <= NW / 2;
    constexpr int MTIP = MTI > 2 ? 4 : (MTI > 1 ? 2 : 1);   // (ESPLIT) input tiles rounded up to a power of two
    // Depthwise / project side: wave -> (output tile wt, chunk pair kp).  KSPLIT = 2 (shapes with at most four
    // output tiles per band): two waves share a tile, each takes one 32-channel chunk pair of every slice and
    // accumulates its own partial project sums; the epilogue adds the two.  MTOW: tiles rounded up to a power of two.
    constexpr int MTOW = MTO > 4 ? 8 : (MTO > 2 ? 4 : (MTO > 1 ? 2 : 1));
    static_assert(KSPLIT == 1 || KSPLIT == 2, "chunk pairs of a slice");
    // (KSPLIT == 1) a band with more output tiles than waves gives wave w the tiles w, w + NW, ...: one after the other
    // in the depthwise + project stage, each with its own project accumulators
    constexpr int MTOT = KSPLIT == 1 ? (MTO + NW - 1) / NW : 1;
    static_assert(KSPLIT == 1 || KSPLIT * MTOW <= NW, "one (output tile, chunk pair) per wave");
    constexpr int TSTEP = NW * RPT;                        // output rows between two tiles of a wave
    static_assert(!RESID || (S == 1 && CIN == COUT), "shortcut needs matching shapes");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(LDS_BYTES == block_lds_bytes<CIN, CEXP, COUT, H, S, TR, MODE, KSPLIT>(), "the launcher's LDS size is this map's");
    static_assert(NT % (COUT / 4) == 0, "epilogue: a thread keeps one channel group");
    static_assert(EPI_BYTES <= OFF_PAR, "the epilogue staging must not reach the per-channel parameters");
    static_assert(NCH % 2 == 0, "slice 0 of the next image goes to the buffer the last slice does not use");
    static_assert(NCH >= 2, "the weight pipeline below assumes at least two slices");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef BK_PRIO_YOUNG       // experiment: the waves dispatched second on each SIMD at raised issue priority (conv_23 0.2167-0.2181 against 0.2135-0.2148 ms: off)
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    const int r = lane & 31, h = lane >> 5;
    const int band = blockIdx.x;
    const int y0 = band * TR;                              // first output row of the band
    const int row0 = WHOLE ? 0 : ((S == 1) ? y0 - 1 : 2 * y0 - 1);   // image row of the first computed input row

    // ---- weight-slice staging by LDS-DMA: piece k of an image is written by wave k % NW; lane l supplies the
    // 16-byte granule 64 k + l of the padded image from wherever it lives in global memory (a padding granule
    // re-reads its row's last unit).  No registers, no store pass; the wave that issued a piece waits for it
    // (vmcnt) before the barrier that publishes the buffer.
    auto stage_w = [&](int c, int buf) {
        char* l1 = lds + (buf ? OFF_WB1 : OFF_WB0);
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
                __builtin_amdgcn_global_load_lds((GPTR)(g3 + (unsigned)(row * 32 + u * 8)), (LPTR)(l1 + W1_BYTES + k * 1024), 16, 0, 0);
            }
        }
    };
    stage_w(0, 0);

    // per-channel parameters of all slices
    for (int i = tid; i < CEXP / 4; i += NT) {
        reinterpret_cast<f32x4*>(lds + OFF_B1)[i] = reinterpret_cast<const f32x4*>(b1p)[i];
        reinterpret_cast<f32x4*>(lds + OFF_S1)[i] = reinterpret_cast<const f32x4*>(s1p)[i];
        reinterpret_cast<f32x4*>(lds + OFF_B2)[i] = reinterpret_cast<const f32x4*>(b2p)[i];
        reinterpret_cast<f32x4*>(lds + OFF_S2)[i] = reinterpret_cast<const f32x4*>(s2p)[i];
    }
    for (int i = tid; i < 3 * CEXP; i += NT) {             // 3x3 weight image: window row, position -> its two words
        const int row = i / CEXP, pos = i - row * CEXP;
        const uint16_t* w2 = reinterpret_cast<const uint16_t*>(W2p) + 3 * row * CEXP + pos;
        char* dst = lds + OFF_W2 + row * dw_wrow(CEXP) + pos * 4;
        *reinterpret_cast<uint32_t*>(dst) = (uint32_t)w2[0] | ((uint32_t)w2[CEXP] << 16);
        *reinterpret_cast<uint32_t*>(dst + dw_wplane(CEXP)) = (uint32_t)w2[2 * CEXP];
    }

    // ---- input fragments (B operand of the expand GEMM), resident while an image's slices are expanded -------
    V8 xf[MTIW][KS1];
    int in_lds[MTIW];     // LDS byte address of this lane's input pixel slot (+ 16 h), -1 = nothing to write
    int x_off[MTIW];      // element offset of this lane's input pixel inside an image (+ 8 h)
    const int we = ESPLIT ? (wave & (MTIP - 1)) : wave, en = ESPLIT ? wave / MTIP : 0;   // wave-uniform
#pragma unroll
    for (int t = 0; t < MTIW; ++t) {
        const int p = 32 * (we + NW * t) + r;
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

    // ---- output pixels of this lane on the depthwise / project side: pixel m of each 16-pixel subtile ----------
    const int m16 = lane & 15, g4 = lane >> 4;
    const int wt = KSPLIT == 1 ? wave : (wave & (MTOW - 1)), kp = KSPLIT == 1 ? 0 : wave / MTOW;   // wave-uniform
    const bool dw_active = wt < MTO && kp < KSPLIT;        // (tile 0; a wave without tile 0 has no later tile either)
    // Subtile 1 holds the pixels 16 lanes further on in the tile: 16 columns further (one tile row of 32), the next
    // row (16) or two rows further (8), always a fixed number of bytes after subtile 0's.  Lanes without an
    // output pixel still read -- their own (unused) column and row, so that the 16 lanes of a read group keep
    // distinct banks; past the end of a row that is the next row, past the last row the weight buffers.
    // Whatever such a lane reads stays in ITS column of the MFMA results (a pixel is a column of both the
    // depthwise and the project product), and that column is never stored.
    constexpr int ST1 = POW == 32 ? 16 * S * SLOT : (16 / POW) * S * PITCH * SLOT;
    static_assert(ST1 + 96 < 65536, "subtile offset must fit the instruction's offset field");
    static_assert(ST1 + 2 * PITCH * SLOT + 112 < 65536, "tap offsets must fit the instruction's offset field");
    int tb;               // LDS byte address of this lane's tap (0, min(g, 2)) in tile 0 (see DwStage::tb); tile t: + t TSTEP S rows
    bool out_ok[MTOT][2];
    int po0;              // this lane's pixel of subtile 0 of tile 0 inside the band; subtile 1 is PO1 further, tile t is t TSTEP rows further
    constexpr int PO1 = POW == 32 ? 16 : (16 / POW) * Wo;
    {
        const int oy = wt * RPT + m16 / POW, ox = m16 % POW;
        po0 = oy * Wo + ox;
#pragma unroll
        for (int t = 0; t < MTOT; ++t) {
            out_ok[t][0] = (ox < Wo) && (oy + t * TSTEP < TR);
            out_ok[t][1] = POW == 32 ? (ox + 16 < Wo) && (oy + t * TSTEP < TR) : (ox < Wo) && (oy + t * TSTEP + 16 / POW < TR);
        }
        tb = ((oy * S) * PITCH + ox * S + (g4 < 2 ? g4 : 2)) * SLOT;
    }
    // project accumulators: tile n of subtile st, register i <-> output channel 16 n + 4 g + i; the bias joins in the epilogue
    f32x4 pacc[MTOT][2][NT16];

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
                        o[4 * a + i] = (T)fmaf(neg_part(v), ex.sl[q][i], v);   // PReLU: sl = slope - 1
                    }
                }
                // 16-byte unit (2 nt + s2) * 2 + h of the slot
                *reinterpret_cast<V8*>(lds + eoff + in_lds[t] + (2 * nt + s2) * 32) = o;
            }
        }
    };
    auto expand = [&](int c, int wb, int eb) {
        if (we >= MTI || en >= 2) return;                  // wave-uniform; such a wave has no later tile either
        const int eoff = eb * ES_BYTES;
        Ex ex;   // lives only inside this call, so its registers are free for the other stage
        ex.w1l = (wb ? OFF_WB1 : OFF_WB0) + r * PW1 + 16 * h;
        ex.pl = OFF_B1 + 256 * c + 32 * h;
        if (!ESPLIT || en == 0) {
            ex.template issue<0>();
            expand_tile(ex, 0, std::true_type{}, 0, eoff);
#pragma unroll
            for (int t = 1; t < MTIW; ++t)
                if (we + NW * t < MTI) expand_tile(ex, t, std::false_type{}, 0, eoff);
        }
        if (!ESPLIT || en == 1) {
            ex.template issue<1>();
            expand_tile(ex, 0, std::true_type{}, 1, eoff);
#pragma unroll
            for (int t = 1; t < MTIW; ++t)
                if (we + NW * t < MTI) expand_tile(ex, t, std::false_type{}, 1, eoff);
        }
    };

    // ================= depthwise 3x3 of slice c from the LDS image -> project MFMA =================================
    auto dwproject = [&](int c, int wb, int eb) {
        if (!dw_active) return;                            // wave-uniform
        DwStage<T, CEXP, NT16, PW3, ST1, PITCH * SLOT, 4 / KSPLIT> dw;
        dw.set_lane(m16, g4);   // rebuilt per slice from the lane index: registers that need not live through the expand stage
        // a wave that takes chunk pair kp of the slice starts 2 kp chunks further in every per-chunk array
        dw.w2l = OFF_W2 + (g4 & 1) * dw_wplane(CEXP) + (64 * c + 32 * kp + m16) * 4;
        dw.pb = OFF_B2 + 256 * c + 128 * kp + 16 * g4;
        dw.w3l = (wb ? OFF_WB1 : OFF_WB0) + W1_BYTES + (16 * (g4 & 1) + m16) * PW3 + 16 * (g4 >> 1) + 32 * kp;
#pragma unroll
        for (int t = 0; t < MTOT; ++t) {
            if (t > 0 && wt + NW * t >= MTO) break;        // wave-uniform
            dw.tb = tb + eb * ES_BYTES + 64 * kp + t * (TSTEP * S * PITCH * SLOT);
            dw.preamble();
            dw.run(pacc[t]);
        }
    };

    // A workgroup keeps its band and walks over images blockIdx.y, blockIdx.y + gridDim.y, ...: the per-channel
    // parameters are staged once, and the next image's input fragments and first weight slice are
    // requested while the current image's last slice and epilogue still run.
    for (int img = blockIdx.y; img < B; img += gridDim.y) {
        const bool has_next = img + (int)gridDim.y < B;
        // zero what the depthwise stage reads of the activation image but the expand stage never writes: the border
        // slots 0 and W + 1 of every row and the rows outside the image (the previous image's epilogue used this
        // memory as staging).  Every other slot a valid lane reads is written in full by each slice's expand.
        {
            constexpr int UB = 2 * 8, UR = PITCH * (SLOT / 16);   // 16-byte units: both border slots of a row; a whole row
            for (int i = tid; i < RIN * UB; i += NT) {
                const int ri = i / UB, u = i - ri * UB;
                *reinterpret_cast<f32x4*>(lds + (ri * PITCH + (u >> 3) * (W + 1)) * SLOT + (u & 7) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // band-uniform: LDS row 0 is image row row0 - ROFF, the last LDS row is image row row0 - ROFF + RIN - 1
            if (row0 - ROFF < 0)
                for (int i = tid; i < UR; i += NT) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row0 - ROFF + RIN - 1 >= H)
                for (int i = tid; i < UR; i += NT)
                    reinterpret_cast<f32x4*>(lds + (RIN - 1) * PITCH * SLOT)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // first image: this wave's pieces of slice 0, requested above.  Later images: they were waited for at the end of
        // the previous image's last slice, and waiting here would only drain that image's output stores.
        if (img == (int)blockIdx.y) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MTOT; ++t)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int n = 0; n < NT16; ++n) pacc[t][st][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // per slice: request the next slice's weights (or the next image's first slice) into the other buffer |
        // expand | barrier | depthwise + project | wait for the requested pieces | barrier
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) stage_w(c + 1, (c + 1) & 1);  // that buffer was last read before the previous barrier
            else if (has_next) stage_w(0, 0);
            expand(c, c & 1, 0);
            if (c == NCH - 1 && has_next) load_x(img + gridDim.y);   // the fragments are dead after the last expand
            // the image of slice c is complete.  Not __syncthreads(): its fence would also wait for the LDS-DMA in flight
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            dwproject(c, c & 1, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                           // image and old weights free, new weights visible
        }

        // ---- epilogue: accumulators -> f32 pixel rows in LDS -> (+ bias, + shortcut) -> NHWC, 8 bytes per lane -------
        // A thread's units (4 channels of one pixel) are NT apart, i.e. a fixed number of pixels: one base address each
        // for the shortcut, the staging rows and the output, everything else is a compile-time offset.  The shortcut
        // values are requested a chunk of EPI_CH units ahead (with the present shapes: all of a thread's units), the
        // first chunk before the accumulators are staged, so their latency runs under the staging pass and its barrier.
        constexpr int G = COUT / 4, UNITS = POUT * G;
        constexpr int EPI_IT = (UNITS + NT - 1) / NT, EPI_CH = EPI_CHUNK, EPI_NCH = (EPI_IT + EPI_CH - 1) / EPI_CH;
        constexpr int PSTEP = NT / G;                          // pixels between two units of a thread
        const int etid = tid;
        const int g = etid % G, p0 = etid / G;
        const size_t pix0 = (size_t)img * Ho * Wo + (size_t)y0 * Wo;
        const T* xs = X + pix0 * CIN;                          // wave-uniform; only dereferenced when RESID
        const unsigned xo = p0 * CIN + 4 * g, oo = p0 * COUT + 4 * g;
        V4 rr[2][EPI_CH];
        auto fetch_shortcut = [&](auto chunk) {
            constexpr int k = decltype(chunk)::value;
            if constexpr (RESID && k < EPI_NCH) {
    #pragma unroll
                for (int j = 0; j < EPI_CH; ++j) {
                    const int it = k * EPI_CH + j;
                    if (it < EPI_IT && p0 + it * PSTEP < POUT) rr[k & 1][j] = *reinterpret_cast<const V4*>(xs + (xo + it * PSTEP * CIN));
                }
            }
        };
        fetch_shortcut(std::integral_constant<int, 0>{});
        if (dw_active) {
    #pragma unroll
            for (int t = 0; t < MTOT; ++t)
    #pragma unroll
                for (int st = 0; st < 2; ++st)
                    if (out_ok[t][st]) {
                        char* dst = lds + kp * EPI1 + (po0 + t * (TSTEP * Wo) + st * PO1) * P32 + 16 * g4;
    #pragma unroll
                        for (int n = 0; n < NT16; ++n) *reinterpret_cast<f32x4*>(dst + 64 * n) = pacc[t][st][n];
                    }
        }
        __syncthreads();
        {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(b3 + 4 * g);
            T* os = O + pix0 * COUT;
            const char* ls = lds + p0 * P32 + g * 16;
            static_for<EPI_NCH>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                fetch_shortcut(std::integral_constant<int, k + 1>{});
    #pragma unroll
                for (int j = 0; j < EPI_CH; ++j) {
                    const int it = k * EPI_CH + j;
                    if (it < EPI_IT && p0 + it * PSTEP < POUT) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(ls + it * PSTEP * P32);
                        if constexpr (KSPLIT == 2) {   // the other chunk pairs' partial sums
                            const f32x4 v2 = *reinterpret_cast<const f32x4*>(ls + EPI1 + it * PSTEP * P32);
    #pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] += v2[i];
                        }
                        if (RESID) {
    #pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] += (float)rr[k & 1][j][i];
                        }
                        V4 o;
    #pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = (T)(v[i] + bias[i]);
                        *reinterpret_cast<V4*>(os + (oo + it * PSTEP * COUT)) = o;
                    }
                }
            });
        }
        if (has_next) __syncthreads();   // staging reads are done before the region is zeroed and restaged
    }
}

// ---------------------------------------------------------------------------------------------
template <int CIN, int CEXP, int COUT, int H, int S, int TR, int MODE, int KSPLIT>
constexpr int block_lds_bytes() {
    // mirrors the LDS map of block_kernel
    constexpr int RIN = (S == 1) ? TR + 2 : 2 * TR + 1;
    constexpr int es = RIN * block_pitch<H, S>() * 144;
    constexpr int wb = 64 * (CIN * 2 + 16) + COUT * 160;
    constexpr int epi = KSPLIT * TR * (H / S) * (COUT * 4 + 16);
    constexpr int off_par = (cmax(es + wb, epi) + 15) / 16 * 16;
    return off_par + 4 * CEXP * 4 + dw_wimage_bytes(CEXP) + wb;
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

template <typename T, int CIN, int CEXP, int COUT, int H, int S, int TR, bool RESID, int NW, int MINW, int MODE, bool PERSIST, int KSPLIT>
static int launch_block_cfg(const BlockParams& p, const void* x, void* out, int B, hipStream_t s) {
    FRA_REQUIRE(p.cout == COUT, FRA_ERR_ARG, "block: cout %d, fused kernel has %d", p.cout, COUT);
    // PERSIST: one workgroup per band and resident slot, each walking over the images blockIdx.y,
    // blockIdx.y + gridDim.y, ... in equal shares (the smallest group count that needs only
    // ceil(B / slots) images per workgroup).  Otherwise one workgroup per band and image, which balances
    // better when the bands do not divide the resident slots evenly.
    const int bands = (H / S) / TR;
    constexpr int lds = block_lds_bytes<CIN, CEXP, COUT, H, S, TR, MODE, KSPLIT>();
    auto kern = block_kernel<T, CIN, CEXP, COUT, H, S, TR, RESID, NW, MINW, MODE, KSPLIT>;
    static std::atomic<uint64_t> lds_set;      // per instantiation: devices whose LDS limit is raised
    static std::atomic<int> slots_of[64];      // resident workgroups per band, per device (0 = not asked yet)
    int dev = 0;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set, &dev)) return rc;
    int slots = dev < 64 ? slots_of[dev].load(std::memory_order_relaxed) : 0;
    if (slots == 0) {
        slots = resident_workgroups(reinterpret_cast<const void*>(kern), NW * 64, lds) / bands;
        if (slots < 1) slots = 1;
        if (dev < 64) slots_of[dev].store(slots, std::memory_order_relaxed);
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

// compute units of the current device (asked once per device)
static int compute_units() {
    static std::atomic<int> cus_of[64];
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev < 64 && (cus = cus_of[dev].load(std::memory_order_relaxed)) > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (dev < 64) cus_of[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

// Two band heights per shape (three for conv_3).  The tall one has the least halo work and is what a full batch runs; with only a
// few images it would leave most of the chip idle (conv_4 as whole-image bands: ONE workgroup per image), so
// a batch whose SHORT bands still fit the chip in one go (one workgroup per compute unit) runs those: more
// redundant expand rows, but 7 to 14 workgroups per image side by side.
template <typename T>
static int launch_block_t(const BlockParams& p, const void* x, void* out, int B, hipStream_t s) {
    const int key = p.cin * 1000000 + p.cexp * 1000 + p.h * 10 + p.stride;
    const int short_bands = (p.h / p.stride) >= 28 ? 14 : 7;   // of the short-band configurations below
    const bool few = (int64_t)B * short_bands <= compute_units();
    switch (key) {
        case 64 * 1000000 + 128 * 1000 + 56 * 10 + 2:   // conv_23
            return few ? launch_block_cfg<T, 64, 128, 64, 56, 2, 2, false, 8, 2, 1, true, 2>(p, x, out, B, s)
                       : launch_block_cfg<T, 64, 128, 64, 56, 2, 4, false, 8, 2, 1, true, 2>(p, x, out, B, s);
        case 64 * 1000000 + 128 * 1000 + 28 * 10 + 1: { // conv_3.*
            if (few) return launch_block_cfg<T, 64, 128, 64, 28, 1, 2, true, 8, 2, 1, true, 1>(p, x, out, B, s);
            // 14-row bands (two output tiles per wave, half the halo rows and half the per-band phases of 7-row
            // bands) cost 1.9 times a 7-row band: take whichever needs less time for its rounds over the chip
            const int cus = compute_units();
            const int64_t cost7 = ((int64_t)4 * B + cus - 1) / cus * 10, cost14 = ((int64_t)2 * B + cus - 1) / cus * 19;
            return cost14 < cost7 ? launch_block_cfg<T, 64, 128, 64, 28, 1, 14, true, 8, 2, 1, true, 1>(p, x, out, B, s)
                                  : launch_block_cfg<T, 64, 128, 64, 28, 1, 7, true, 8, 2, 1, true, 1>(p, x, out, B, s);
        }
        case 64 * 1000000 + 256 * 1000 + 28 * 10 + 2:   // conv_34
            return few ? launch_block_cfg<T, 64, 256, 128, 28, 2, 2, false, 8, 2, 1, true, 2>(p, x, out, B, s)
                       : launch_block_cfg<T, 64, 256, 128, 28, 2, 7, false, 8, 2, 1, true, 2>(p, x, out, B, s);
        // conv_4.* (128 -> 256 -> 128 on 14 x 14, residual) is launch_rblock's (kernels_rblock.hip)
        case 128 * 1000000 + 512 * 1000 + 14 * 10 + 2:  // conv_45
            return few ? launch_block_cfg<T, 128, 512, 128, 14, 2, 1, false, 8, 2, 1, true, 2>(p, x, out, B, s)
                       : launch_block_cfg<T, 128, 512, 128, 14, 2, 7, false, 8, 2, 1, true, 2>(p, x, out, B, s);
        case 128 * 1000000 + 256 * 1000 + 7 * 10 + 1:   // conv_5.*
            return few ? launch_block_cfg<T, 128, 256, 128, 7, 1, 1, true, 8, 2, 1, true, 2>(p, x, out, B, s)
                       : launch_block_cfg<T, 128, 256, 128, 7, 1, 7, true, 8, 2, 1, true, 2>(p, x, out, B, s);
        default: break;
    }
    FRA_REQUIRE(false, FRA_ERR_ARG, "block: no fused kernel for cin=%d cexp=%d h=%d stride=%d", p.cin, p.cexp, p.h,
                p.stride);
    return FRA_ERR_ARG;
}

// conv_3 runs as the row pipeline of kernels_pblock.hip where that is the faster one (FRA_PBLOCK_MIN=<batch>: from that
// batch on instead, 0: never -- for measurements)
bool block_uses_pipeline(const BlockParams& p, int B) {
    static const int pb_min = [] { const char* e = getenv("FRA_PBLOCK_MIN"); return e ? atoi(e) : -1; }();
    return pb_min < 0 ? pblock_pays(p, B) : (pb_min > 0 && pblock_has(p) && B >= pb_min);
}

int launch_block(int dt, const BlockParams& p, const void* x, void* out, int B, hipStream_t s) {
    FRA_REQUIRE(x && out && p.w1 && p.b1p && p.s1p && p.w2p && p.b2p && p.s2p && p.w3p && p.b3, FRA_ERR_ARG,
                "block: null pointer");
    FRA_REQUIRE(B > 0 && B <= 65535, FRA_ERR_ARG, "block: batch %d not in [1,65535]", B);
    const bool want_res = p.residual != 0;
    const bool is_res_cfg = (p.stride == 1);
    FRA_REQUIRE(want_res == is_res_cfg, FRA_ERR_ARG, "block: stride %d with residual=%d has no fused kernel", p.stride,
                p.residual);
    if (block_uses_pipeline(p, B)) return launch_pblock(dt, p, x, out, B, s);
    if (dt == ACT_F16) return launch_block_t<f16>(p, x, out, B, s);
    if (dt == ACT_BF16) return launch_block_t<bf16>(p, x, out, B, s);
    FRA_REQUIRE(false, FRA_ERR_ARG, "block: unsupported activation dtype %d", dt);
    return FRA_ERR_ARG;
}

}  // namespace fra
