// Fused network head for gfx950: Conv2d(3,64,3x3,s2)+BN+PReLU (face_model.py:80) followed by the
// depthwise 3x3 Conv_block (face_model.py:81) in one kernel, reading the caller's crops directly:
// either the reference's fp32 NCHW tensor in [-1,1] or raw uint8 HWC BGR pixels, in which case
// ToTensor + Normalize(0.5, 0.5) (facebank.py:22-24) is applied while staging.  Images whose index
// is >= n_src are the horizontally MIRRORED copies of image (index - n_src): that is the second
// forward of the flip test-time augmentation (facebank.py:59-61) without a second input tensor.
//
// Same structure as the fused block (kernels_block.hip): a workgroup owns TR output rows of one
// image.  It stages the input rows it needs (3 channel planes, 16-bit, zero padded) in LDS with
// coalesced loads, runs the stem conv on the matrix cores as an im2col GEMM (K = 27 taps padded to
// 32; a lane gathers its 16 taps from the LDS planes), writes PReLU(stem) into an LDS image
// [row][x+1][64 ch] with zero borders, and every lane then computes the depthwise 3x3 for its pixel
// from nine 16-byte LDS reads per 8 channels and stores NHWC.  The 56x56x64 stem output never
// touches HBM.
//
// The stem weight matrix is stored with its rows (output channels) permuted by perm_pos(): MFMA
// row 16g+8a+4h+i computes output channel 16g+8h+4a+i, so the 8 values a lane holds per
// 16-channel group are 8 CONSECUTIVE channels and travel as one 16-byte unit.  Slope arrays hold
// (PReLU slope - 1).
//
// LDS image: rows of SD_ROWB bytes, pixel slots of 128 bytes (slot = x + 1; slots 0 and 57 are the zero border); unit u
// (8 channels) of slot s sits at u ^ (s & 7).  The stem stage writes it from the MFMA layout (lane = pixel, two units
// per lane), which the XOR keeps conflict free.
//
// The depthwise stage runs on the SPARSE 16x16x64 MFMA like the fused blocks' (block_stages.h has the operand maps):
// D[ch][px] = sum over (tap slot, ch') Diag[ch][(slot, ch')] . E[(slot, ch')][px], one instruction per window row
// (tap slots 0..2 = taps (p, 0..2), slot 3 has weight zero), for 16 channels and 16 pixels.  Wave w owns columns
// 8 w .. 8 w + 7 of the band's four rows as two tiles of 2 rows x 8 columns: tile t = rows t and t + 2.  A B lane
// (pixel m, tap slot g) reads the two units of a 16-channel chunk of ITS tap column for image rows a + 2 (m >> 3):
// that operand ("set a") is window row a of tile 0 AND window row a - 1 of tile 1, so the four sets of a chunk feed
// six MFMAs -- 32 LDS reads per lane and band where one lane per (pixel, unit) needs 36.  The XOR key depends on the
// column only, so every tap address is one of eight lane registers plus an immediate; a row pitch of 58.5 slots puts
// rows two apart into opposite halves of the LDS banks and the 16 pixels of a set read 16 different bank groups.
// The diagonal operands of the 12 (window row, chunk) pairs are built once per workgroup and kept in LDS as ready
// fragments (16 bytes per lane); the accumulators start from the bias.  The reads run SD_PD sets ahead of the MFMAs through a register ring with
// counted waits (LDS returns data in order).  A lane ends up with channels 16 c + 4 g .. + 3 of its pixel; one
// v_permlane16_swap per register between the results of chunks 2 h and 2 h + 1 leaves every lane with one whole
// unit (16 bytes) and the four lanes of a pixel with 64 contiguous bytes of the NHWC output per store instruction.
#include "common.h"

namespace fra {

#define STEM_LDS_READ16(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

constexpr int SD_ROWB = 58 * 128 + 64;       // bytes per image row in LDS
#ifndef SD_PD_SETS
#define SD_PD_SETS 2
#endif
constexpr int SD_PD = SD_PD_SETS, SD_RING = SD_PD + 1;   // sets of LDS reads in flight ahead of the MFMAs

// depthwise stage of one wave: 4 output rows x 8 columns x 64 channels
template <typename T> struct StemDwStage {
    typedef typename vec4<T>::type V4;
    typedef typename vec8<T>::type V8;
    typedef typename vec16<T>::type V16;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    static constexpr int SETS = 16;       // (chunk c, set a): step 4 c + a
    V8 er[SD_RING][2];
    V8 dg[SD_RING + 1];                   // the diagonal of (window row a, chunk c) arrives with set (c, a); tile 1 uses it one set later
    int idx;                              // positions of the sparse fragment's elements
    f32x4 slope[4];                       // channels 16 c + 4 g .. + 3; slope - 1
    f32x4 acc[2][2];                      // [chunk parity][tile]: the bias of chunk c arrives in them with set (c, 0)
    int pb;                               // LDS address of the bias quad of channels 4 g ..; + 64 c
    int dga;                              // LDS address of this lane's diagonal fragment (0, 0); + ((4 p + c) * 64) * 16
    u32x2 held[2];                        // the even chunk's rounded results, per tile
    int ta[8];                            // LDS address of unit u of this lane's tap column, set 0
    T* out;                               // this lane's 16 output bytes of tile 0, chunk pair 0

    // m = lane & 15, g = lane >> 4; the same fragment maps as DwStage::set_lane
    static __device__ __forceinline__ void lane_masks(int m, int g, uint32_t (&msk)[4], int& idx) {
        const bool mine = (m >> 3) == (g >> 1);
        const int r = (m & 7) >> 2, pos = m & 3, other = pos ^ 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) msk[i] = !mine ? 0u : (i == r ? 0x0000FFFFu : (i == 2 + r ? 0xFFFF0000u : 0u));
        idx = 0x4444;
        idx = (idx & ~(0xF << (4 * r))) | ((pos | (other << 2)) << (4 * r));
        idx = (idx & ~(0xF << (4 * (2 + r)))) | ((other | (pos << 2)) << (4 * (2 + r)));
    }
    // one lane's diagonal fragment of (window row p, chunk c) from the [9][64] depthwise weights: row m of the
    // chunk, tap slots 2 (g & 1) and 2 (g & 1) + 1 of unit g >> 1 (slot 3 has weight zero)
    static __device__ __forceinline__ u32x4 fragment(const unsigned short* w2, int p, int c, int m, int g) {
        uint32_t msk[4];
        int idx_unused;
        lane_masks(m, g, msk, idx_unused);
        const uint32_t even = w2[(3 * p + 2 * (g & 1)) * 64 + 16 * c + m];
        const uint32_t odd = w2[(3 * p + 1) * 64 + 16 * c + m];
        const uint32_t w = even | ((g & 1) ? 0u : odd << 16);
        return u32x4{w & msk[0], w & msk[1], w & msk[2], w & msk[3]};
    }
    static constexpr int reads_of(int S) { return S % 4 == 0 ? 5 : S % 4 < 3 ? 3 : 2; }
    static constexpr int reads_after(int S) {        // LDS reads issued after set S's when the wave waits for it
        int n = 0;
        for (int k = S + 1; k <= S + SD_PD && k < SETS; ++k) n += reads_of(k);
        return n;
    }
    template <int S> __device__ __forceinline__ void issue() {
        constexpr int c = S / 4, a = S % 4;
        STEM_LDS_READ16(er[S % SD_RING][0], ta[2 * c], a * SD_ROWB);
        STEM_LDS_READ16(er[S % SD_RING][1], ta[2 * c + 1], a * SD_ROWB);
        if constexpr (a < 3) STEM_LDS_READ16(dg[S % (SD_RING + 1)], dga, ((4 * a + c) * 64) * 16);
        if constexpr (a == 0) {
            STEM_LDS_READ16(acc[c & 1][0], pb, 64 * c);
            STEM_LDS_READ16(acc[c & 1][1], pb, 64 * c);
        }
    }
    template <int S = 0> __device__ __forceinline__ void preamble() {
        if constexpr (S < SD_PD) { issue<S>(); preamble<S + 1>(); }
    }
    // PReLU + rounding of a finished tile; odd chunks pair up with the even chunk before them and store
    template <int C, int TILE> __device__ __forceinline__ void finish() {
        V4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (T)fmaf(neg_part(acc[C & 1][TILE][i]), slope[C][i], acc[C & 1][TILE][i]);
        const u32x2 w = __builtin_bit_cast(u32x2, o);
        if constexpr (C % 2 == 0) {
            held[TILE] = w;
        } else {
            // rows (16 lanes) g = 0, 2 keep the even chunk's values and fetch their right neighbour's (the next four
            // channels); rows g = 1, 3 fetch the odd chunk's values of their left neighbour and keep their own
            const u32x2 s0 = __builtin_amdgcn_permlane16_swap(held[TILE][0], w[0], false, false);
            const u32x2 s1 = __builtin_amdgcn_permlane16_swap(held[TILE][1], w[1], false, false);
            const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
            *reinterpret_cast<u32x4*>(out + (size_t)TILE * 56 * 64 + 32 * (C / 2)) = v;
        }
    }
    template <int S = 0> __device__ __forceinline__ void run() {
        if constexpr (S < SETS) {
            constexpr int c = S / 4, a = S % 4, slot = S % SD_RING, ds = S % (SD_RING + 1), dp = (S + SD_RING) % (SD_RING + 1);
            if constexpr (S + SD_PD < SETS) issue<S + SD_PD>();
            if constexpr (a == 0)
                asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(er[slot][0]), "+v"(er[slot][1]), "+v"(dg[ds]), "+v"(acc[c & 1][0]), "+v"(acc[c & 1][1]) : "n"(reads_after(S)));
            else if constexpr (a < 3)
                asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(er[slot][0]), "+v"(er[slot][1]), "+v"(dg[ds]) : "n"(reads_after(S)));
            else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(er[slot][0]), "+v"(er[slot][1]) : "n"(reads_after(S)));
            const V16 e = __builtin_shufflevector(er[slot][0], er[slot][1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
            if constexpr (a <= 2) acc[c & 1][0] = smfma16(dg[ds], e, acc[c & 1][0], idx);
            if constexpr (a >= 1) acc[c & 1][1] = smfma16(dg[dp], e, acc[c & 1][1], idx);     // window row a - 1: the previous set's diagonal
            if constexpr (a == 3) {
                finish<c, 0>();
                finish<c, 1>();
            }
            __builtin_amdgcn_sched_barrier(0);
            run<S + 1>();
        }
    }
};

template <typename T, bool U8, int TR, int NW, int MINW>
__global__ __launch_bounds__(NW * 64, MINW)
void stemdw_kernel(const void* __restrict__ Xv, int n_src, int n_items, const T* __restrict__ WA,
                   const float* __restrict__ b1, const float* __restrict__ s1, const T* __restrict__ W2,
                   const float* __restrict__ b2, const float* __restrict__ s2, T* __restrict__ O) {
    typedef typename vec8<T>::type V8;
    typedef typename vec4<T>::type V4;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    constexpr int NT = NW * 64;
    constexpr int H = 56, W = 56, HI = 112, NB = H / TR;
    constexpr int RIN = TR + 2;
    constexpr int PIN = RIN * W;
    static_assert(PIN % 16 == 0, "stem stage: whole tiles of 16 pixels");
    constexpr int MTI = PIN / 16;
    static_assert(W % 8 == 0 && W / 8 <= NW && SD_ROWB >= (W + 2) * 128 && TR == 4, "depthwise stage: one wave per 8-pixel column group, two tiles of two rows");
    constexpr int ES_BYTES = RIN * SD_ROWB;
    // staged input: 4 planes (3 channels + one all-zero plane for the 5 padding taps) of NR rows x PW columns;
    // input column x sits at column x + 4, so groups of 4 columns line up with 16-byte source loads
    constexpr int NR = 2 * TR + 5, PW = 120, NG = PW / 4;
    constexpr int OFF_IN = ES_BYTES, IN_BYTES = 4 * NR * PW * 2;
    constexpr int OFF_W2 = OFF_IN + IN_BYTES, OFF_B1 = OFF_W2 + 9 * 64 * 2, OFF_S1 = OFF_B1 + 256, OFF_B2 = OFF_S1 + 256,
                  OFF_S2 = OFF_B2 + 256, OFF_WA = OFF_S2 + 256, OFF_DG = OFF_WA + 64 * 32 * 2,
                  OFF_TOFF = OFF_DG + 12 * 64 * 16;
    static_assert(H % TR == 0, "band height");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const es = lds;
    T* const inp = reinterpret_cast<T*>(lds + OFF_IN);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // A workgroup walks over items (image, band of TR output rows): item, item + gridDim.x, ...  The input rows of
    // the NEXT item are requested into registers before this item's stem stage and written to the LDS planes right
    // after it (the planes are free then, and the depthwise stage needs the registers), so a workgroup waits a
    // whole memory latency once, at its first item.
    // Input band: column group g holds input columns 4(g-1) .. 4(g-1)+3 (g = 0 and g = NG-1 are padding).
    // staging pieces: 4 pixels of one row (x 3 channels for U8).  Half a wave (32 lanes, 30 used) takes one row of
    // column groups, NT / 32 rows per pass: shifts and masks instead of divisions by 30 and 13
    constexpr int NROWS = U8 ? NR : 3 * NR, ROWS_PER_PASS = NT / 32;
    constexpr int IN_IT = (NROWS + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
    static_assert(NG <= 32, "staging map: one row of column groups per half wave");
    f32x4 vin[U8 ? 1 : IN_IT];
    unsigned win[U8 ? IN_IT : 1][3];
    // (sg, scr: this thread's column group and first row; rebuilt from an opaque copy of the thread index wherever
    // they are used, so that nothing derived from them stays in registers across the compute stages)
    auto piece_ok = [&](int sg, int scr, int it, int band) {
        const int cr = scr + ROWS_PER_PASS * it;                        // cr = c * NR + ry (fp32) or ry (uint8)
        const int c = U8 ? 0 : (cr >= NR) + (cr >= 2 * NR), ry = cr - c * NR;
        const int yin = 2 * (band * TR - 1) - 1 + ry;                   // LDS input row 0 holds input row 2 row0 - 1
        return cr < NROWS && sg >= 1 && sg <= 28 && yin >= 0 && yin < HI;
    };
    auto request = [&](int item) {
        int ot = tid;
        asm volatile("" : "+v"(ot));
        const int sg = ot & 31, scr = ot >> 5;
        const int img = item / NB, band = item - img * NB;
        const int src = img >= n_src ? img - n_src : img;
        const bool flip = img >= n_src;
        const int x0 = flip ? 108 - 4 * (sg - 1) : 4 * (sg - 1);
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int cr = scr + ROWS_PER_PASS * it;
            const int c = U8 ? 0 : (cr >= NR) + (cr >= 2 * NR), ry = cr - c * NR;
            const int yin = 2 * (band * TR - 1) - 1 + ry;
            const bool ok = piece_ok(sg, scr, it, band);
            if (!U8) {
                vin[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok)
                    vin[it] = *reinterpret_cast<const f32x4*>(static_cast<const float*>(Xv) + (size_t)src * 3 * HI * HI +
                                                              ((size_t)c * HI + yin) * HI + x0);
            } else {
                win[it][0] = win[it][1] = win[it][2] = 0;
                if (ok) {   // uint8 HWC: 4 pixels x 3 channels = 12 contiguous bytes
                    const unsigned* pw = reinterpret_cast<const unsigned*>(static_cast<const unsigned char*>(Xv) +
                                                                          (size_t)src * HI * HI * 3 + ((size_t)yin * HI + x0) * 3);
                    win[it][0] = pw[0]; win[it][1] = pw[1]; win[it][2] = pw[2];
                }
            }
        }
    };
    // the requested rows: registers -> 16-bit LDS planes
    auto stage = [&](int item) {
        int ot = tid;
        asm volatile("" : "+v"(ot));
        const int sg = ot & 31, scr = ot >> 5;
        const int img = item / NB, band = item - img * NB;
        const bool flip = img >= n_src;
#pragma unroll
        for (int it = 0; it < IN_IT; ++it) {
            const int cr = scr + ROWS_PER_PASS * it;
            if (cr >= NROWS || sg >= NG) continue;
            if (!U8) {
                f32x4 v = vin[it];
                if (flip) v = f32x4{v[3], v[2], v[1], v[0]};
                V4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (T)v[j];
                *reinterpret_cast<V4*>(inp + cr * PW + 4 * sg) = o;
            } else {
                // ToTensor (/255) then Normalize ((t - 0.5) / 0.5)
                const bool ok = piece_ok(sg, scr, it, band);
                const unsigned long long lo = (unsigned long long)win[it][0] | ((unsigned long long)win[it][1] << 32);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    V4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int byte = 3 * j + c;          // pixel j, channel c
                        const unsigned u = byte < 8 ? (unsigned)(lo >> (8 * byte)) & 255u : (win[it][2] >> (8 * (byte - 8))) & 255u;
                        // ((u / 255) - 0.5) / 0.5 as one FMA: after rounding to the 16-bit storage type it gives the
                        // reference's value for every one of the 256 inputs (checked exhaustively, f16 and bf16)
                        const float f = ok ? fmaf((float)u, 2.0f / 255.0f, -1.0f) : 0.0f;
                        o[flip ? 3 - j : j] = (T)f;
                    }
                    *reinterpret_cast<V4*>(inp + (c * NR + cr) * PW + 4 * sg) = o;
                }
            }
        }
    };
    int item = blockIdx.x;
    request(item);

    // ---- once per workgroup: weights and parameters to LDS, the zero parts of both images
    {
        V8 wav = {};
        static_assert(64 * 32 / 8 <= NT, "one thread per 16-byte piece of the stem weights");
        if (tid < 64 * 32 / 8) wav = reinterpret_cast<const V8*>(WA)[tid];
        f32x4 pv[4] = {};
        if (tid < 16) {
            pv[0] = reinterpret_cast<const f32x4*>(b1)[tid];
            pv[1] = reinterpret_cast<const f32x4*>(s1)[tid];
            pv[2] = reinterpret_cast<const f32x4*>(b2)[tid];
            pv[3] = reinterpret_cast<const f32x4*>(s2)[tid];
        }
        // the border slots 0 and W + 1 of every row of the activation image (the stem stage never writes them) and
        // the padding plane of the input
        for (int i = tid; i < RIN * 16; i += NT) {
            const int ri = i >> 4, u = i & 15;
            *reinterpret_cast<f32x4*>(es + ri * SD_ROWB + (u >> 3) * (W + 1) * 128 + (u & 7) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int i = tid; i < NR * PW * 2 / 16; i += NT)
            reinterpret_cast<f32x4*>(lds + OFF_IN + 3 * NR * PW * 2)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the stem stage's gather table: tap t -> plane c = t / 9 (3 = zero plane), ky, kx; element offset from a
        // lane's patch origin (LDS row 2 iy, column 2 ix + 3).  Lane group g gathers taps 8 g .. 8 g + 7
        if (tid < 32) {
            const int t = tid, c = t / 9, rem = t - 9 * c, ky = rem / 3, kx = rem - 3 * ky;
            reinterpret_cast<int*>(lds + OFF_TOFF)[tid] = (c * NR + ky) * PW + kx + 3;
        }
        // the depthwise stage's diagonal fragments: [4 p + c][lane]
        for (int i = tid; i < 12 * 64; i += NT) {
            const int pc = i >> 6, l = i & 63;
            *reinterpret_cast<typename StemDwStage<T>::u32x4*>(lds + OFF_DG + i * 16) =
                StemDwStage<T>::fragment(reinterpret_cast<const unsigned short*>(W2), pc >> 2, pc & 3, l & 15, l >> 4);
        }
        if (tid < 64 * 32 / 8) reinterpret_cast<V8*>(lds + OFF_WA)[tid] = wav;
        if (tid < 16) {
            reinterpret_cast<f32x4*>(lds + OFF_B1)[tid] = pv[0];
            reinterpret_cast<f32x4*>(lds + OFF_S1)[tid] = pv[1];
            reinterpret_cast<f32x4*>(lds + OFF_B2)[tid] = pv[2];
            reinterpret_cast<f32x4*>(lds + OFF_S2)[tid] = pv[3];
        }
    }

    stage(item);
#pragma unroll 1
    for (; item < n_items; item += gridDim.x) {
        const int img = item / NB, band = item - img * NB;
        const int y0 = band * TR, row0 = y0 - 1;

        __syncthreads();
        if (item + (int)gridDim.x < n_items) request(item + gridDim.x);      // wave-uniform; lands during the stem stage
        // the first band's row above the image and the last band's row below it: zeros, which the stem stage (it
        // skips pixels outside the image) leaves alone; the previous item's depthwise stage is done with the row
        if (row0 < 0)
            for (int i = tid; i < SD_ROWB / 16; i += NT) reinterpret_cast<f32x4*>(es)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row0 + RIN > H)
            for (int i = tid; i < SD_ROWB / 16; i += NT)
                reinterpret_cast<f32x4*>(es + (RIN - 1) * SD_ROWB)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ================= stem: im2col GEMM -> PReLU -> LDS image =================
        // 16x16x32 MFMAs: one instruction covers the 27 taps (K padded to 32) for 16 output channels and 16 pixels.
        // Weight row 16 n + 4 g + i holds channel 8 (4 (n >> 1) + g) + 4 (n & 1) + i, so a lane's results of row
        // tiles 2 v and 2 v + 1 are the 8 consecutive channels of unit 4 v + g.  21 tiles of 16 pixels over 8 waves.
        {
            // (lane constants are rebuilt per item from an opaque copy of the lane index: kept live across the loop
            // they would not fit the register budget of the depthwise stage)
            int m = lane & 15, g = lane >> 4;
            asm volatile("" : "+v"(m), "+v"(g));
            V8 wa[4];
            f32x4 bq[4], sq[4];             // bias and slope - 1 of the lane's channels, per row tile
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                wa[n] = *reinterpret_cast<const V8*>(lds + OFF_WA + ((16 * n + m) * 32 + 8 * g) * 2);
                bq[n] = *reinterpret_cast<const f32x4*>(lds + OFF_B1 + (8 * (4 * (n >> 1) + g) + 4 * (n & 1)) * 4);
                sq[n] = *reinterpret_cast<const f32x4*>(lds + OFF_S1 + (8 * (4 * (n >> 1) + g) + 4 * (n & 1)) * 4);
            }
            int toff[8];                    // the 8 taps this lane gathers, as element offsets from its patch origin
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const i32x4 v = *reinterpret_cast<const i32x4*>(lds + OFF_TOFF + (8 * g + 4 * q) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) toff[4 * q + i] = v[i];
            }
#pragma unroll 1
            for (int tt = wave; tt < MTI; tt += NW) {
                const int p = 16 * tt + m;
                const int iy = p / W, ix = p - iy * W;
                const int gy = row0 + iy;
                const bool ok = (gy >= 0) && (gy < H);          // rows outside the image hold zeros already
                const T* patch = inp + (2 * iy) * PW + 2 * ix;
                V8 bfr;
#pragma unroll
                for (int j = 0; j < 8; ++j) bfr[j] = patch[toff[j]];
                const int sx = ix + 1;                 // slot of the pixel in its image row
                char* const px = es + iy * SD_ROWB + sx * 128;
                f32x4 acc[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = mfma16(wa[n], bfr, bq[n]);
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    V8 o;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float x = acc[2 * v + a][i];
                            o[4 * a + i] = (T)fmaf(neg_part(x), sq[2 * v + a][i], x);   // PReLU: sq = slope - 1
                        }
                    const int u = 4 * v + g;       // channels [8u, 8u+8)
                    if (ok) *reinterpret_cast<V8*>(px + ((u ^ (sx & 7)) << 4)) = o;
                }
            }
        }
        __syncthreads();
        if (item + (int)gridDim.x < n_items) stage(item + gridDim.x);        // the input planes are free again
        __builtin_amdgcn_sched_barrier(0);    // the depthwise stage needs the registers the staged rows held

        // ================= depthwise 3x3 + PReLU from the LDS image -> NHWC =================
        if (wave < W / 8) {                       // wave-uniform; wave w owns output columns 8 w .. 8 w + 7 of every row
            StemDwStage<T> dw;
            int m = lane & 15, g = lane >> 4;
            asm volatile("" : "+v"(m), "+v"(g));
            {
                uint32_t msk_unused[4];
                StemDwStage<T>::lane_masks(m, g, msk_unused, dw.idx);
            }
            dw.dga = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + OFF_DG + (16 * g + m) * 16;
            dw.pb = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + OFF_B2 + 16 * g;
            // B side: pixel m of tile t = (row t + 2 (m >> 3), column 8 w + (m & 7)); tap slot g = tap column
            // min(g, 2); output column x reads slots x .. x + 2
            const int sx = 8 * wave + (m & 7) + (g < 2 ? g : 2);
            const int tap = (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + 2 * (m >> 3) * SD_ROWB + sx * 128;
#pragma unroll
            for (int u = 0; u < 8; ++u) dw.ta[u] = tap + ((u ^ (sx & 7)) << 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dw.slope[c] = *reinterpret_cast<const f32x4*>(lds + OFF_S2 + (16 * c + 4 * g) * 4);
            }
            // after the lane swap: rows g = 0, 2 hold unit 4 h + (g >> 1), rows g = 1, 3 unit 4 h + 2 + (g >> 1)
            dw.out = O + ((size_t)img * H * W + (size_t)(y0 + 2 * (m >> 3)) * W + 8 * wave + (m & 7)) * 64 + 16 * (g & 1) + 8 * (g >> 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dw.preamble();
            dw.run();
        }
    }
}

template <typename T, bool U8, int TR, int NW, int MINW>
static int launch_stemdw_cfg(const StemDwParams& p, const void* x, int n_src, void* out, int B, hipStream_t s) {
    constexpr int lds = (TR + 2) * SD_ROWB + 4 * (2 * TR + 5) * 120 * 2 + 9 * 64 * 2 + 4 * 256 + 64 * 32 * 2 + 12 * 64 * 16 + 128;
    auto kern = stemdw_kernel<T, U8, TR, NW, MINW>;
    static std::atomic<uint64_t> lds_set;   // per instantiation: devices whose LDS limit is raised
    static std::atomic<int> slots_of[64];   // resident workgroups, per device (0 = not asked yet)
    int dev = 0;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set, &dev)) return rc;
    int slots = dev < 64 ? slots_of[dev].load(std::memory_order_relaxed) : 0;
    if (slots == 0) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), NW * 64, (size_t)lds) != hipSuccess || per_cu <= 0)
            per_cu = 1;
        slots = cus * per_cu;
        if (dev < 64) slots_of[dev].store(slots, std::memory_order_relaxed);
    }
    // one workgroup per resident slot, each walking over its (image, band) items in equal shares
    const int items = B * (56 / TR);
    const int per = (items + slots - 1) / slots;
    const int groups = (items + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3(groups), dim3(NW * 64), lds, s, x, n_src, items, (const T*)p.wa, p.b1, p.s1,
                       (const T*)p.w2, p.b2, p.s2, (T*)out);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

int launch_stemdw(int dt, const StemDwParams& p, const void* x, int x_is_u8, int n_src, void* out, int B,
                  hipStream_t s) {
    FRA_REQUIRE(x && out && p.wa && p.b1 && p.s1 && p.w2 && p.b2 && p.s2, FRA_ERR_ARG, "stemdw: null pointer");
    FRA_REQUIRE(B > 0 && B <= 65535, FRA_ERR_ARG, "stemdw: batch %d not in [1,65535]", B);
    FRA_REQUIRE(n_src > 0 && n_src <= B && B <= 2 * n_src, FRA_ERR_ARG, "stemdw: %d images from %d sources", B, n_src);
    if (dt == ACT_F16)
        return x_is_u8 ? launch_stemdw_cfg<f16, true, 4, 8, 4>(p, x, n_src, out, B, s)
                       : launch_stemdw_cfg<f16, false, 4, 8, 4>(p, x, n_src, out, B, s);
    if (dt == ACT_BF16)
        return x_is_u8 ? launch_stemdw_cfg<bf16, true, 4, 8, 4>(p, x, n_src, out, B, s)
                       : launch_stemdw_cfg<bf16, false, 4, 8, 4>(p, x, n_src, out, B, s);
    FRA_REQUIRE(false, FRA_ERR_ARG, "stemdw: unsupported activation dtype %d", dt);
    return FRA_ERR_ARG;
}

}  // namespace fra
