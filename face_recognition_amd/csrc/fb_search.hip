// Facebank matching: squared-L2 nearest neighbours of query embeddings in a gallery.
//
// Two storage modes:
//   * fp32 gallery  -- exact evaluation of sum_d (q_d - g_d)^2, the arithmetic of the reference's
//     demos (cam_demo.py:97-99); meant for the small facebanks those demos use.
//   * 16-bit gallery -- score(q, g) = q.g - |g|^2/2 computed on the matrix cores with the query
//     on the MFMA column (lane) axis, so every lane owns ONE query and keeps that query's running
//     top-k in registers while gallery tiles stream through LDS.  |q|^2 - 2 score is the squared
//     distance.  Per-lane lists are merged by a small selection kernel.
//
// Ordering everywhere: ascending distance, equal distances -> lower index first.
#include <math.h>
#include <stdlib.h>

#include "common.h"

#ifndef SEARCH_WIDE_DEFAULT
#define SEARCH_WIDE_DEFAULT 0
#endif

struct fb_handle {
    int device = 0;
    int64_t N = 0;      // rows
    int64_t Npad = 0;   // rows allocated (a whole number of search tiles for the packed layout)
    int D = 0;
    int store = 0;      // FRA_F32 / FRA_BF16 / FRA_F16
    int64_t index_base = 0;
    void* rows = nullptr;      // f32 [N][D] or 16-bit [Npad][D]
    float* negh = nullptr;     // -|g|^2/2 per row (16-bit modes), -inf in padding rows
    // scratch, grown on demand.  A block that has been handed to kernels is never freed before
    // fb_destroy: a captured hipGraph may still hold its address.
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    void* retired[32] = {nullptr};
    int n_retired = 0;
    hipStream_t copy_stream = nullptr;   // fb_create / fb_fill pack rows here, so they never synchronise other streams
};

namespace fra {

static const int SEARCH_D = 512;   // feature width of the MFMA search kernel
static const int BQ = 256;         // queries per block (8 waves x 32)
static const int BG = 48;          // gallery rows per LDS tile

// ------------------------------------------------------------------------------------------------
// pack: f32 rows -> 16-bit rows + (-|g|^2/2) of the ROUNDED values.  One wave per row.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                        float* __restrict__ negh, long n, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float ss = 0.0f;
    for (int d = lane; d < D; d += 64) {
        const T v = (T)src[row * D + d];
        dst[row * D + d] = v;
        const float f = (float)v;
        ss = fmaf(f, f, ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (lane == 0) negh[row] = -0.5f * ss;
}

template <typename T>
__global__ __launch_bounds__(256) void pad_rows_kernel(T* __restrict__ dst, float* __restrict__ negh, long n0,
                                                       long n1, int D) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (n1 - n0) * D;
    if (i < total) dst[n0 * D + i] = (T)0.0f;
    if (i < n1 - n0) negh[n0 + i] = -INFINITY;
}

// queries: f32 [Q][D] -> 16-bit [Qpad][D] (zero rows beyond Q) and |q|^2 of the f32 values
template <typename T>
__global__ __launch_bounds__(256) void prep_queries_kernel(const float* __restrict__ q, T* __restrict__ qh,
                                                           float* __restrict__ qn, int Q, int Qpad, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Qpad) return;
    float ss = 0.0f;
    for (int d = lane; d < D; d += 64) {
        const float f = row < Q ? q[(size_t)row * D + d] : 0.0f;
        qh[(size_t)row * D + d] = (T)f;
        ss = fmaf(f, f, ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (lane == 0) qn[row] = ss;
}

// ------------------------------------------------------------------------------------------------
// MFMA search.  Block = 8 waves; wave w owns queries [q0 + 32 w, +32) as two 16-query tiles whose
// 16x16x32 B fragments stay in registers for the whole kernel (16 k-steps x 2 tiles x 4 VGPRs).
// Gallery tiles of 48 rows (48 KB, plus their 48 bias terms) arrive in LDS by direct global->LDS
// loads through a ring of three buffers, two tiles ahead of the one being consumed; the 16-byte
// chunk c of row r sits at chunk position c ^ (r & 15), applied on the global source address
// because the LDS side of such a load is lane-linear.
//
// A tile is consumed as three chains of 16 rows; chain rt accumulates rows 16 rt .. +16 against both
// query tiles over the 16 k-steps.  Lane (col = lane & 15, kg = lane >> 4) then holds, for query
// col of each tile, the scores of rows 16 rt + 4 kg + i (i < 4), and keeps a running top-KS list per
// query tile for that quarter of the rows.  The list update of a chain is issued inside the next
// chain (three accumulator sets rotate), so its vector work overlaps matrix work.
//
// The LDS reads and their waits are written by hand: reads run PD steps ahead of the MFMAs through a
// ring of NB register slots -- across tile boundaries too -- and since LDS returns data in order,
// "at most n reads outstanding" (s_waitcnt lgkmcnt(n)) is exactly "the fragment of this step has
// arrived".  The one barrier per tile sits PD steps before the tile's end, just ahead of the first
// read of the next tile, so the matrix pipeline never drains at a tile boundary.  The buffer a
// tile occupies is overwritten by loads that any wave issues only after that barrier plus several
// steps, when every read of the tile has long been issued and returned.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* GPTR;  // global source of an LDS-DMA load
typedef __attribute__((address_space(3))) void* LPTR;        // its (wave-uniform) LDS destination

template <int KS> struct TopList {
    float s[KS];
    int id[KS];
    // empty slots carry id -1 and the score `floor`: only rows that score above it are ever taken in
    __device__ __forceinline__ void init(float floor) {
#pragma unroll
        for (int j = 0; j < KS; ++j) { s[j] = floor; id[j] = -1; }
    }
    // rows reach a lane in increasing order, so on equal scores the resident (lower) row stays ahead
    __device__ __forceinline__ void insert(float v, int row) {
        if (v > s[KS - 1]) { s[KS - 1] = v; id[KS - 1] = row; }
#pragma unroll
        for (int j = KS - 1; j > 0; --j) {
            const bool up = s[j] > s[j - 1];
            const float ts = s[j - 1]; const int ti = id[j - 1];
            s[j - 1] = up ? s[j] : ts;   id[j - 1] = up ? id[j] : ti;
            s[j] = up ? ts : s[j];       id[j] = up ? ti : id[j];
        }
    }
};

static const int SEARCH_KSTEPS = SEARCH_D / 32;                  // k-steps of one chain
static const int SEARCH_CHAINS = BG / 16;                        // chains of one tile (3)
static const int SEARCH_STEPS = SEARCH_CHAINS * SEARCH_KSTEPS;   // steps of one tile
static const int SEARCH_TILE_BYTES = BG * SEARCH_D * 2;          // 49152
static const int SEARCH_BUF_BYTES = SEARCH_TILE_BYTES + 256;     // + bias terms (a 64-lane load fills 256 B)

// The block's 256 queries are dealt to its waves QTW tiles of 16 each: QTW = 2 is eight waves, two to a SIMD; QTW = 4 is
// FOUR waves, one to a SIMD, each with the whole 512-entry register file: every gallery fragment read from LDS then
// feeds four matrix instructions instead of two (half the LDS reads per multiply-add), the 256 registers of query
// fragments sit in the accumulation half of the file (the MFMAs are inline assembly with an "a" operand for them) and
// one in-order stream of four MFMAs per step keeps the matrix pipe busy without a partner wave.
template <int QTW> struct SearchGeom {
    static constexpr int NWV = BQ / (16 * QTW);                  // waves of a block
    static constexpr int PIECES = BG / NWV;                      // 1 KB rows each wave stages per tile
    static constexpr int PIECE_EVERY = SEARCH_STEPS / PIECES;    // a piece goes out every so many steps ...
    static constexpr int PIECE_AT = PIECE_EVERY / 2 - 1;         // ... at this step of each group
    static constexpr int NSRC = 16 / NWV;                        // distinct swizzles (row & 15) among a wave's rows
    static_assert(BQ % (16 * QTW) == 0 && BG % NWV == 0 && SEARCH_STEPS % PIECES == 0 && 16 % NWV == 0, "geometry");
};

// one MFMA of the search: gallery fragment a (rows), query fragment b (columns)
template <int QTW> struct SearchMma {
    // eight waves: the compiler's own instruction
    template <typename V8> static __device__ __forceinline__ void first(f32x4& d, V8 a, V8 b, f32x4 c) { d = mfma16(a, b, c); }
    template <typename V8> static __device__ __forceinline__ void next(f32x4& d, V8 a, V8 b) { d = mfma16(a, b, d); }
};
template <> struct SearchMma<4> {
    // four wide waves: the query fragment comes from an accumulation register
    static __device__ __forceinline__ void first(f32x4& d, vec8<f16>::type a, vec8<f16>::type b, f32x4 c) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    }
    static __device__ __forceinline__ void next(f32x4& d, vec8<f16>::type a, vec8<f16>::type b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    }
    static __device__ __forceinline__ void first(f32x4& d, vec8<bf16>::type a, vec8<bf16>::type b, f32x4 c) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    }
    static __device__ __forceinline__ void next(f32x4& d, vec8<bf16>::type a, vec8<bf16>::type b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    }
};

// NB = A-fragment register slots (a divisor of SEARCH_STEPS so the ring carries across tiles);
// reads run PD = NB - 1 steps ahead
template <typename T, int KS, int NB, int QTW> struct SearchCtx {
    typedef typename vec8<T>::type V8;
    V8 qf[SEARCH_KSTEPS][QTW];     // query fragments
    V8 af[NB];                     // ring of gallery fragments
    f32x4 acc[SEARCH_CHAINS][QTW]; // [chain][query tile]
    f32x4 bias;                    // -|g|^2/2 of the chain's 4 rows of this lane
    TopList<KS> top[QTW];          // per query tile
    int bl[4], bn[4];              // lane part of the fragment address for k-step & 3: this tile, next tile
    int nha, nhn;                  // LDS address of this lane's bias terms of chain 0: this tile, next tile
    int kg;
    // staging of the tile two ahead
    const T* gsrc[SearchGeom<QTW>::NSRC];   // this lane's source of piece p: gsrc[p % NSRC] (the swizzle follows row & 15)
    const float* nsrc;             // bias source (wave 0)
    char* ldst;                    // LDS destination of piece 0
    int wave;
    bool do_stage;
};

// fold chain `rt` of the tile whose first row is row0 into the lists
template <typename Ctx, int KS, int QTW>
__device__ __forceinline__ void fold_chain(Ctx& c, int row0, int rt) {
    bool any = false;
#pragma unroll
    for (int u = 0; u < QTW; ++u) {
        const float m = fmaxf(fmaxf(c.acc[rt][u][0], c.acc[rt][u][1]), fmaxf(c.acc[rt][u][2], c.acc[rt][u][3]));
        any = any || m > c.top[u].s[KS - 1];
    }
    if (__any(any)) {
        const int rbase = row0 + 16 * rt + 4 * c.kg;
#pragma unroll
        for (int u = 0; u < QTW; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = c.acc[rt][u][i];
                if (v > c.top[u].s[KS - 1]) c.top[u].insert(v, rbase + i);
            }
    }
}

// -DSEARCH_PRIO=n (experiments): issue priority inside a step.  1: raised while a wave requests its staging piece;
// 2: raised for the waves dispatched second (4-7) throughout;  3: raised around the matrix instructions of a step;
// 4: raised during the list update.  10k queries x 1M / 8M rows, one box, ms: 0: 8.20-8.21 / 58.7-58.8,  1: 8.25 / 58.9,
// 2: 8.16 / 58.6,  3: 8.36 / 59.7,  4: 8.21 / 58.7.  2 against 0 three times each on another box: 8.105-8.149 against
// 8.169-8.194 ms at 1M rows, 115.35-115.89 against 115.72-116.23 ms at 16M: 0.3-0.5 % every time (the waves dispatched
// second otherwise get the matrix pipe only when the first ones leave it), adopted
#ifndef SEARCH_PRIO
#define SEARCH_PRIO 2
#endif
template <typename T, int KS, int NB, int QTW, int S> struct SearchSteps {
    typedef typename vec8<T>::type V8;
    typedef SearchCtx<T, KS, NB, QTW> Ctx;
    typedef SearchGeom<QTW> GEO;
    typedef SearchMma<QTW> MMA;
    static constexpr int PD = NB - 1;
    static constexpr int BARRIER_STEP = SEARCH_STEPS - PD - 1;   // the step after which the next tile is first read
    // pieces go out at steps PIECE_EVERY p + PIECE_AT: those issued up to the barrier step
    static constexpr int PIECES_BEFORE_BARRIER = (BARRIER_STEP - GEO::PIECE_AT) / GEO::PIECE_EVERY + 1;
    static_assert(SEARCH_STEPS % NB == 0, "ring slots must line up across tiles");
    static_assert(GEO::PIECE_EVERY * (GEO::PIECES - 1) + GEO::PIECE_AT < SEARCH_STEPS && GEO::PIECE_AT >= 0, "every piece needs a step");
    // reads issued after the fragment of step ss and before its wait: the PD fragments that follow
    // (into the next tile if need be) plus the bias reads that precede a chain's first fragment
    static constexpr int ahead(int ss) {
        int n = PD;
        for (int m = SEARCH_KSTEPS; m <= SEARCH_STEPS; m += SEARCH_KSTEPS)
            if (ss < m && m <= ss + PD) ++n;
        return n;
    }
    // LIN = step index counted from the start of this tile; LIN >= STEPS addresses the next tile
    template <int LIN> static __device__ __forceinline__ void read_step(Ctx& c) {
        constexpr bool nxt = LIN >= SEARCH_STEPS;
        constexpr int SS = nxt ? LIN - SEARCH_STEPS : LIN;
        constexpr int rt = SS / SEARCH_KSTEPS, ks = SS % SEARCH_KSTEPS;
        if constexpr (ks == 0)   // the chain's bias terms go first: they are needed with its first fragment
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c.bias) : "v"(nxt ? c.nhn : c.nha), "n"(rt * 64));
        asm volatile("ds_read_b128 %0, %1 offset:%2"
                     : "=v"(c.af[LIN % NB])
                     : "v"(nxt ? c.bn[ks & 3] : c.bl[ks & 3]), "n"((ks >> 2) * 256 + rt * 16384));
    }
    static __device__ __forceinline__ void prologue(Ctx& c) {
        if constexpr (S < PD) {
            read_step<S>(c);
            SearchSteps<T, KS, NB, QTW, S + 1>::prologue(c);
        }
    }
    // one piece (a 1 KB row) of the tile two ahead; wave w stages rows w, w + NWV, ...
    template <int PIECE> static __device__ __forceinline__ void stage_piece(Ctx& c) {
        if (c.do_stage) {
            constexpr int row8 = PIECE * GEO::NWV;
            __builtin_amdgcn_global_load_lds((GPTR)(c.gsrc[PIECE % GEO::NSRC] + (size_t)row8 * SEARCH_D),
                                             (LPTR)(c.ldst + row8 * (SEARCH_D * 2)), 16, 0, 0);
            if (PIECE == GEO::PIECES - 1 && c.wave == 0)   // wave 0: ldst is the buffer base
                __builtin_amdgcn_global_load_lds((GPTR)c.nsrc, (LPTR)(c.ldst + SEARCH_TILE_BYTES), 4, 0, 0);
        }
    }
    static __device__ __forceinline__ void run(Ctx& c, int row0, bool have_prev) {
        if constexpr (S < SEARCH_STEPS) {
            constexpr int rt = S / SEARCH_KSTEPS, ks = S % SEARCH_KSTEPS, slot = S % NB;
            read_step<S + PD>(c);
            if (SEARCH_PRIO == 3) __builtin_amdgcn_s_setprio(1);
            if constexpr (ks == 0) {
                // wait for this step's fragment and the chain's bias; both are named so that their
                // consumers cannot be moved above the wait
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(c.af[slot]), "+v"(c.bias) : "n"(ahead(S)));
#pragma unroll
                for (int u = 0; u < QTW; ++u) MMA::first(c.acc[rt][u], c.af[slot], c.qf[0][u], c.bias);
            } else {
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(c.af[slot]) : "n"(ahead(S)));
#pragma unroll
                for (int u = 0; u < QTW; ++u) MMA::next(c.acc[rt][u], c.af[slot], c.qf[ks][u]);
            }
            if (SEARCH_PRIO == 3) __builtin_amdgcn_s_setprio(0);
            if constexpr (S % GEO::PIECE_EVERY == GEO::PIECE_AT && S / GEO::PIECE_EVERY < GEO::PIECES) {
                if (SEARCH_PRIO == 1) __builtin_amdgcn_s_setprio(1);
                stage_piece<S / GEO::PIECE_EVERY>(c);
                if (SEARCH_PRIO == 1) __builtin_amdgcn_s_setprio(0);
            }
            if constexpr (ks == 2) {
                if (SEARCH_PRIO == 4) __builtin_amdgcn_s_setprio(1);   // the previous chain finished two steps ago: fold it in under this chain
                if constexpr (rt == 0) { if (have_prev) fold_chain<Ctx, KS, QTW>(c, row0 - BG, SEARCH_CHAINS - 1); }
                else fold_chain<Ctx, KS, QTW>(c, row0, rt - 1);
                if (SEARCH_PRIO == 4) __builtin_amdgcn_s_setprio(0);
            }
            if constexpr (S == BARRIER_STEP) {
                // tile barrier, just before the first read of the next tile: this wave's pieces of the
                // next tile (issued one tile ago) have landed once only this tile's pieces remain
                if (c.do_stage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES_BEFORE_BARRIER));
                else asm volatile("s_waitcnt vmcnt(0)");
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            SearchSteps<T, KS, NB, QTW, S + 1>::run(c, row0, have_prev);
        }
    }
};

template <typename T, int KS, int QTW>
__device__ __forceinline__ void search_body(const T* __restrict__ G, const float* __restrict__ negh,
                                            const T* __restrict__ Qh, const float* __restrict__ qn,
                                            float* __restrict__ pdist, int* __restrict__ pidx,
                                            const float* __restrict__ qfloor, int tile0,
                                            int tiles_total, int tiles_per_split, int QT, int S, int Qpad,
                                            int index_base) {
    typedef typename vec8<T>::type V8;
    typedef SearchGeom<QTW> GEO;
    constexpr int D = SEARCH_D, CH = D / 8;
    constexpr int NB = (KS > 5 && QTW == 2) ? 6 : 8;   // eight waves: the longer top-k lists leave room for fewer fragment slots
    typedef SearchCtx<T, KS, NB, QTW> Ctx;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kg = lane >> 4;

    // block -> (query tile, gallery split); blocks that share blockIdx % 8 share an XCD's L2, so
    // let them walk the same gallery range with different query tiles
    int qt, split;
    {
        const int b = blockIdx.x;
        if ((S & 7) == 0) { const int j = b >> 3; split = (b & 7) + 8 * (j / QT); qt = j % QT; }
        else { qt = b % QT; split = b / QT; }
    }
    const int t_begin = tile0 + split * tiles_per_split;
    int t_end = t_begin + tiles_per_split;
    if (t_end > tiles_total) t_end = tiles_total;

    Ctx c;
    c.kg = kg;
    c.wave = wave;
    // query fragments (MFMA B operand): lane (col, kg) holds Q[q][32 s + 8 kg .. +8] of query col of tile u
    const int qrow0 = qt * BQ + wave * (16 * QTW) + col;
#pragma unroll
    for (int s = 0; s < SEARCH_KSTEPS; ++s)
#pragma unroll
        for (int u = 0; u < QTW; ++u)
            c.qf[s][u] = *reinterpret_cast<const V8*>(Qh + (size_t)(qrow0 + 16 * u) * D + s * 32 + kg * 8);
    // qfloor: per query a score that its k-th best row is known to beat (from a pass over other rows); the lists
    // then skip everything below it from their first row on instead of filling up with whatever comes first
#pragma unroll
    for (int u = 0; u < QTW; ++u) c.top[u].init(qfloor ? qfloor[qrow0 + 16 * u] : -INFINITY);

    // A fragment of step (rt, ks): row 16 rt + col, chunk 4 ks + kg  ->  lane part lo[ks & 3] plus the
    // compile-time part 256 (ks >> 2) + 16384 rt
    const int lds0 = (int)(unsigned)(size_t)(LPTR)smem;
    int lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) lo[j] = col * (CH * 16) + (((4 * j + kg) ^ col) << 4);

    // staging: piece p of a tile is row NWV p + wave, and lane l fetches the chunk that belongs at
    // position l of that row: l ^ (row & 15), which takes NSRC values over a wave's pieces
    int chv[GEO::NSRC];
#pragma unroll
    for (int j = 0; j < GEO::NSRC; ++j) chv[j] = lane ^ ((GEO::NWV * j + wave) & 15);
    auto stage_all = [&](int tile, int bufsel) {
        char* base = smem + bufsel * SEARCH_BUF_BYTES;
        const T* gt = G + (size_t)tile * BG * D;
#pragma unroll
        for (int i = 0; i < GEO::PIECES; ++i) {
            const int row = i * GEO::NWV + wave;
            __builtin_amdgcn_global_load_lds((GPTR)(gt + (size_t)row * D + chv[i % GEO::NSRC] * 8),
                                             (LPTR)(base + row * (CH * 16)), 16, 0, 0);
        }
        if (wave == 0)   // lanes past the tile's rows re-read row 0 into the padding of the bias area
            __builtin_amdgcn_global_load_lds((GPTR)(negh + (size_t)tile * BG + (lane < BG ? lane : 0)),
                                             (LPTR)(base + SEARCH_TILE_BYTES), 4, 0, 0);
    };

    if (t_begin < t_end) {
        stage_all(t_begin, 0);
        if (t_begin + 1 < t_end) stage_all(t_begin + 1, 1);
    }
    __syncthreads();   // drains the loads above

    if (SEARCH_PRIO == 2 && wave >= GEO::NWV / 2) __builtin_amdgcn_s_setprio(1);
    if (t_begin < t_end) {
        int cur = 0;   // buffer of tile t; t + 1 sits in cur + 1, t + 2 is staged into cur + 2 (mod 3)
#pragma unroll
        for (int j = 0; j < 4; ++j) c.bl[j] = lo[j] + lds0;
        c.nha = lds0 + SEARCH_TILE_BYTES + 16 * kg;
        __builtin_amdgcn_sched_barrier(0);
        SearchSteps<T, KS, NB, QTW, 0>::prologue(c);
        for (int t = t_begin; t < t_end; ++t) {
            const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
            // the reads that run ahead into the next tile are issued for the last tile as well (from
            // this tile's buffer, results unused) so that the counted waits stay exact
            const int rb = (t + 1 < t_end ? nxt : cur) * SEARCH_BUF_BYTES + lds0;
#pragma unroll
            for (int j = 0; j < 4; ++j) c.bn[j] = lo[j] + rb;
            c.nhn = rb + SEARCH_TILE_BYTES + 16 * kg;
            c.do_stage = t + 2 < t_end;
            c.ldst = smem + nn * SEARCH_BUF_BYTES + wave * (CH * 16);
            c.nsrc = negh + (size_t)(t + 2) * BG + (lane < BG ? lane : 0);
#pragma unroll
            for (int j = 0; j < GEO::NSRC; ++j) c.gsrc[j] = G + (size_t)(t + 2) * BG * D + (size_t)wave * D + chv[j] * 8;
            __builtin_amdgcn_sched_barrier(0);
            SearchSteps<T, KS, NB, QTW, 0>::run(c, t * BG, t > t_begin);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) c.bl[j] = c.bn[j];
            c.nha = c.nhn;
            cur = nxt;
        }
        // retire the reads issued ahead for a tile that does not exist before their registers are reused
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c.bias));
#pragma unroll
        for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(c.af[i]));
        fold_chain<Ctx, KS, QTW>(c, (t_end - 1) * BG, SEARCH_CHAINS - 1);
    }

    // partial lists: [P = 4 S][Qpad][KS], list p = 4 split + kg
#pragma unroll
    for (int u = 0; u < QTW; ++u) {
        const int qrow = qrow0 + 16 * u;
        const float q2 = qn[qrow];
        const size_t o = ((size_t)(4 * split + kg) * Qpad + qrow) * KS;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const bool ok = c.top[u].id[j] >= 0;
            pdist[o + j] = ok ? fmaxf(q2 - 2.0f * c.top[u].s[j], 0.0f) : INFINITY;
            pidx[o + j] = ok ? c.top[u].id[j] + index_base : -1;
        }
    }
}

template <typename T, int KS>
__global__ __launch_bounds__(512) void search_kernel(const T* __restrict__ G, const float* __restrict__ negh,
                                                     const T* __restrict__ Qh, const float* __restrict__ qn,
                                                     float* __restrict__ pdist, int* __restrict__ pidx,
                                                     const float* __restrict__ qfloor, int tile0,
                                                     int tiles_total, int tiles_per_split, int QT, int S, int Qpad,
                                                     int index_base) {
    search_body<T, KS, 2>(G, negh, Qh, qn, pdist, pidx, qfloor, tile0, tiles_total, tiles_per_split, QT, S, Qpad, index_base);
}
// four wide waves, one per SIMD (SearchGeom above)
template <typename T, int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void search_kernel_wide(const T* __restrict__ G, const float* __restrict__ negh,
                        const T* __restrict__ Qh, const float* __restrict__ qn,
                        float* __restrict__ pdist, int* __restrict__ pidx,
                        const float* __restrict__ qfloor, int tile0,
                        int tiles_total, int tiles_per_split, int QT, int S, int Qpad,
                        int index_base) {
    search_body<T, KS, 4>(G, negh, Qh, qn, pdist, pidx, qfloor, tile0, tiles_total, tiles_per_split, QT, S, Qpad, index_base);
}

// ------------------------------------------------------------------------------------------------
// exact fp32 distances: dist[q][n] = sum_d (q_d - g_d)^2, one wave per (q, n) pair  (cam_demo.py:97-98)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2dist_f32_kernel(const float* __restrict__ q, const float* __restrict__ g,
                                                         float* __restrict__ dist, int Q, long N, int D) {
    const int lane = threadIdx.x & 63;
    const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= (long)Q * N) return;
    const long qi = pair / N, n = pair - qi * N;
    float ss = 0.0f;
    for (int d = lane; d < D; d += 64) {
        const float df = q[qi * D + d] - g[n * D + d];
        ss = fmaf(df, df, ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (lane == 0) dist[pair] = ss;
}

// ------------------------------------------------------------------------------------------------
// selection: for query q pick the k smallest (dist, idx) pairs, lexicographically, out of
// candidates c = (list, j): dist[list*strideL + q*strideQ + j], j < len.  idx == nullptr means the
// candidate's index is j + index_base (dense distance row).  Entries with idx < 0 are empty.
// Round i picks the smallest pair greater than round i-1's pick, so nothing is marked or moved.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pair_less(float d1, int i1, float d2, int i2) {
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

// CACHE > 0: the candidates number at most 256 CACHE and a thread keeps its share in registers over the rounds
// (partial lists of a search, shard lists of a merge); CACHE == 0: every round reads them again (a dense distance row).
static const int SELECT_CACHE = 8;
template <int CACHE>
__global__ __launch_bounds__(256) void select_topk_kernel(const float* __restrict__ dist, const int* __restrict__ idx,
                                                          int lists, long strideL, long strideQ, int len,
                                                          int index_base, int k, float* __restrict__ dist_out,
                                                          int* __restrict__ idx_out) {
    __shared__ float sd[4];
    __shared__ int si[4];
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long total = (long)lists * len;
    float last_d = -INFINITY;
    int last_i = -1;
    float cd[CACHE > 0 ? CACHE : 1];
    int ci[CACHE > 0 ? CACHE : 1];
    if constexpr (CACHE > 0) {
#pragma unroll
        for (int u = 0; u < CACHE; ++u) {
            const long c = tid + 256L * u;
            cd[u] = INFINITY;
            ci[u] = -1;
            if (c < total) {
                const long l = c / len, j = c - l * len;
                const long a = l * strideL + (long)q * strideQ + j;
                cd[u] = dist[a];
                ci[u] = idx ? idx[a] : (int)j + index_base;
            }
        }
    }
    for (int round = 0; round < k; ++round) {
        float bd = INFINITY;
        int bi = 0x7fffffff;
        if constexpr (CACHE > 0) {
#pragma unroll
            for (int u = 0; u < CACHE; ++u) {
                const float d = cd[u];
                const int i = ci[u];
                const bool open = i >= 0 && (round == 0 || pair_less(last_d, last_i, d, i));
                if (open && pair_less(d, i, bd, bi)) { bd = d; bi = i; }
            }
        } else {
            for (long c = tid; c < total; c += 256) {
                const long l = c / len, j = c - l * len;
                const long a = l * strideL + (long)q * strideQ + j;
                const float d = dist[a];
                const int i = idx ? idx[a] : (int)j + index_base;
                if (i < 0) continue;
                if (round > 0 && !pair_less(last_d, last_i, d, i)) continue;
                if (pair_less(d, i, bd, bi)) { bd = d; bi = i; }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(bd, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (pair_less(od, oi, bd, bi)) { bd = od; bi = oi; }
        }
        if (lane == 0) { sd[wave] = bd; si[wave] = bi; }
        __syncthreads();
        bd = sd[0]; bi = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (pair_less(sd[w], si[w], bd, bi)) { bd = sd[w]; bi = si[w]; }
        __syncthreads();
        const bool found = bi != 0x7fffffff;
        if (tid == 0) {
            dist_out[(size_t)q * k + round] = found ? bd : INFINITY;
            idx_out[(size_t)q * k + round] = found ? bi : -1;
        }
        if (!found) {
            // nothing left: fill the remaining slots and stop (uniform across the block)
            if (tid == 0)
                for (int rr = round + 1; rr < k; ++rr) {
                    dist_out[(size_t)q * k + rr] = INFINITY;
                    idx_out[(size_t)q * k + rr] = -1;
                }
            break;
        }
        last_d = bd;
        last_i = bi;
    }
}

// One WAVE per query (four queries per block) for up to 64 WSEL candidates -- the partial lists of a search, the shard
// lists of a merge: a lane keeps its candidates in registers, a round is a lane-local minimum and six shuffle steps,
// nothing goes through LDS and nothing waits at a barrier.  Same order as select_topk_kernel: ascending
// (distance, index), round i picks the smallest pair greater than round i - 1's.
// kth_floor != nullptr: instead of the k results, the score floor of the query (below).
static const int WSEL = 16;
__global__ __launch_bounds__(256) void select_topk_wave_kernel(const float* __restrict__ dist, const int* __restrict__ idx,
                                                               int lists, long strideL, long strideQ, int len, int k, int Q,
                                                               float* __restrict__ dist_out, int* __restrict__ idx_out,
                                                               const float* __restrict__ qn, float* __restrict__ kth_floor) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;                                    // wave-uniform
    const int total = lists * len;
    float cd[WSEL];
    int ci[WSEL];
#pragma unroll
    for (int u = 0; u < WSEL; ++u) {
        const int c = lane + 64 * u;
        cd[u] = INFINITY;
        ci[u] = -1;
        if (c < total) {
            const int l = c / len, j = c - l * len;
            const long a = l * strideL + (long)q * strideQ + j;
            cd[u] = dist[a];
            ci[u] = idx[a];
        }
    }
    float last_d = -INFINITY;
    int last_i = -1;
    bool found = true;
    for (int round = 0; round < k; ++round) {
        float bd = INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < WSEL; ++u) {
            const float d = cd[u];
            const int i = ci[u];
            const bool open = i >= 0 && (round == 0 || pair_less(last_d, last_i, d, i));
            if (open && pair_less(d, i, bd, bi)) { bd = d; bi = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(bd, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (pair_less(od, oi, bd, bi)) { bd = od; bi = oi; }
        }
        found = found && bi != 0x7fffffff;
        if (!kth_floor && lane == 0) {
            dist_out[(size_t)q * k + round] = found ? bd : INFINITY;
            idx_out[(size_t)q * k + round] = found ? bi : -1;
        }
        last_d = found ? bd : INFINITY;
        last_i = found ? bi : 0x7fffffff;                  // nothing is greater than this pair: later rounds find nothing
    }
    if (kth_floor && lane == 0) {
        const float q2 = qn[q];
        kth_floor[q] = found ? 0.5f * (q2 - last_d) - 4e-7f * (fabsf(q2) + last_d) - 1e-30f : -INFINITY;
    }
}

// ------------------------------------------------------------------------------------------------
// Score floor of each query from the partial lists of a pass over PART of the gallery (select_topk_wave_kernel with
// kth_floor): the k-th smallest distance d_k among them bounds the k-th smallest over the whole gallery from above,
// so every row of the final answer scores at least (|q|^2 - d_k) / 2.  The floor is that score minus a margin for the
// two roundings between a score and its distance (rows that TIE with the k-th best must still pass the lists' strict
// comparison).  Fewer than k candidates: no floor.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void threshold_score_kernel(const float* __restrict__ minimum,
                                                              long long* __restrict__ idx,
                                                              float* __restrict__ score, int Q, float limit) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Q) return;
    const float m = minimum[i];
    if (m > limit) idx[i] = -1;                                   // cam_demo.py:100
    // cam_demo.py:105 -- multiply and add round separately in the reference, so no fused multiply-add
    // (plain operators under the pragma: the __fmul_rn / __fadd_rn helpers fuse after inlining)
    float sc;
    {
#pragma clang fp contract(off)
        sc = m * -80.0f + 156.0f;
    }
    score[i] = fminf(fmaxf(sc, 0.0f), 100.0f);
}

// The demos' whole post-processing of one query's distance row in one launch (cam_demo.py:98-105): the nearest row --
// the lexicographically smallest (distance, index) pair, as select_topk_kernel's first round picks it --, its index as
// int64 or -1 beyond the threshold, and the score of threshold_score_kernel.  One block per query.
__global__ __launch_bounds__(256) void match_nearest_kernel(const float* __restrict__ dist, long N, int index_base,
                                                            float limit, float* __restrict__ minimum,
                                                            long long* __restrict__ idx_out, float* __restrict__ score) {
    __shared__ float sd[4];
    __shared__ int si[4];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = dist + (size_t)q * N;
    float bd = INFINITY;
    int bi = 0x7fffffff;
    for (long c = tid; c < N; c += 256) {
        const float d = row[c];
        const int i = (int)c + index_base;
        if (pair_less(d, i, bd, bi)) { bd = d; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float od = __shfl_xor(bd, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (pair_less(od, oi, bd, bi)) { bd = od; bi = oi; }
    }
    if (lane == 0) { sd[wave] = bd; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        bd = sd[0]; bi = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (pair_less(sd[w], si[w], bd, bi)) { bd = sd[w]; bi = si[w]; }
        const bool found = bi != 0x7fffffff;
        const float m = found ? bd : INFINITY;
        minimum[q] = m;
        idx_out[q] = (!found || m > limit) ? -1 : (long long)bi;          // cam_demo.py:100
        float sc;
        {
#pragma clang fp contract(off)
            sc = m * -80.0f + 156.0f;                                      // cam_demo.py:105: two roundings, as there
        }
        score[q] = fminf(fmaxf(sc, 0.0f), 100.0f);
    }
}

// ------------------------------------------------------------------------------------------------
static void launch_select(hipStream_t s, int Q, const float* dist, const int* idx, int lists, long strideL, long strideQ, int len,
                          int index_base, int k, float* dist_out, int* idx_out) {
    if (idx && (long)lists * len <= 64L * WSEL)
        hipLaunchKernelGGL(select_topk_wave_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, dist, idx, lists, strideL, strideQ, len, k, Q,
                           dist_out, idx_out, (const float*)nullptr, (float*)nullptr);
    else if ((long)lists * len <= 256L * SELECT_CACHE)
        hipLaunchKernelGGL(select_topk_kernel<SELECT_CACHE>, dim3(Q), dim3(256), 0, s, dist, idx, lists, strideL, strideQ, len,
                           index_base, k, dist_out, idx_out);
    else
        hipLaunchKernelGGL(select_topk_kernel<0>, dim3(Q), dim3(256), 0, s, dist, idx, lists, strideL, strideQ, len, index_base, k,
                           dist_out, idx_out);
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int ensure_scratch(fb_handle* h, size_t bytes) {
    if (h->scratch_bytes >= bytes) return FRA_OK;
    size_t want = bytes > 2 * h->scratch_bytes ? bytes : 2 * h->scratch_bytes;   // geometric growth bounds the retired list
    void* fresh = nullptr;
    FRA_HIP(fra::dev_malloc(&fresh, want));
    if (h->scratch) {
        if (h->n_retired < 32) {
            h->retired[h->n_retired++] = h->scratch;
        } else {                       // cannot happen with doubling (2^32 growth); drain and free rather than leak
            (void)hipDeviceSynchronize();
            (void)fra::dev_free(h->scratch);
        }
    }
    h->scratch = fresh;
    h->scratch_bytes = want;
    return FRA_OK;
}

template <typename T>
static void launch_pack(const float* src, fb_handle* h, int64_t row0, int64_t m, hipStream_t s) {
    hipLaunchKernelGGL(pack_rows_kernel<T>, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s, src,
                       static_cast<T*>(h->rows) + row0 * h->D, h->negh + row0, (long)m, h->D);
}

// Rows [row0, row0 + n) <- src.  `ordered` != nullptr: device rows, everything is queued on that stream and nothing
// is waited for.  Otherwise the work goes to the handle's own stream, which is synchronised before returning (the
// other streams of the process are left alone); the caller must have finished writing device rows.
static int fill_rows(fb_handle* h, int64_t row0, const float* src, int is_device, int64_t n, const hipStream_t* ordered = nullptr) {
    FRA_REQUIRE(row0 >= 0 && n >= 0 && row0 + n <= h->N, FRA_ERR_ARG, "fb_fill: rows [%lld,%lld) outside [0,%lld)",
                (long long)row0, (long long)(row0 + n), (long long)h->N);
    if (n == 0) return FRA_OK;
    FRA_DEVICE(h->device);
    const int D = h->D;
    if (!ordered && !h->copy_stream) FRA_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    hipStream_t s = ordered ? *ordered : h->copy_stream;
    if (h->store == FRA_F32) {
        FRA_HIP(hipMemcpyAsync(static_cast<float*>(h->rows) + row0 * D, src, (size_t)n * D * 4,
                               is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
        if (!ordered) FRA_HIP(hipStreamSynchronize(s));
        return FRA_OK;
    }
    const int64_t step = is_device ? n : (int64_t)(64 << 20) / (D * 4);  // host data goes through a 64 MB bounce
    float* bounce = nullptr;
    if (!is_device) FRA_HIP(fra::dev_malloc(&bounce, (size_t)step * D * 4));
    int rc = FRA_OK;
    for (int64_t o = 0; o < n && rc == FRA_OK; o += step) {
        const int64_t m = (n - o) < step ? (n - o) : step;
        const float* dsrc = src + o * D;
        if (!is_device) {
            // the previous chunk's pack kernel (same stream) has read the bounce buffer before this copy overwrites it
            hipError_t e = hipMemcpyAsync(bounce, src + o * D, (size_t)m * D * 4, hipMemcpyHostToDevice, s);
            if (e != hipSuccess) { set_error("fb_fill: copy to the device failed: %s", hipGetErrorString(e)); rc = FRA_ERR_HIP; break; }
            dsrc = bounce;
        }
        if (h->store == FRA_F16) launch_pack<f16>(dsrc, h, row0 + o, m, s);
        else launch_pack<bf16>(dsrc, h, row0 + o, m, s);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("fb_fill: pack kernel launch failed: %s", hipGetErrorString(e)); rc = FRA_ERR_HIP; }
    }
    if (!ordered || bounce) {
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess && rc == FRA_OK) { set_error("fb_fill: pack kernel failed: %s", hipGetErrorString(e)); rc = FRA_ERR_HIP; }
    }
    if (bounce) (void)fra::dev_free(bounce);
    return rc;
}

// compute units of the current device (asked once per device)
static int search_compute_units() {
    static std::atomic<int> cus_of[64];
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev < 64 && (cus = cus_of[dev].load(std::memory_order_relaxed)) > 0) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (dev < 64) cus_of[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

// Scratch of one search: `fixed` == nullptr takes (and grows) the handle's own block; otherwise the search must
// fit the `cap` bytes at `fixed` and nothing is allocated.
static int take_scratch(fb_handle* h, size_t need, char* fixed, size_t cap, char** out) {
    if (fixed) {
        FRA_REQUIRE(need <= cap, FRA_ERR_ARG,
                    "fb_search_l2_topk_noalloc: this search needs %zu bytes of scratch, the static array holds %zu; "
                    "use fb_search_l2_topk (fewer queries, or a smaller fp32 gallery, fit)", need, cap);
        *out = fixed;
        return FRA_OK;
    }
    int rc = ensure_scratch(h, need);
    if (rc) return rc;
    *out = static_cast<char*>(h->scratch);
    return FRA_OK;
}

template <typename T, int KSV>
static int launch_search(const dim3 grid, size_t lds, hipStream_t s, const T* G, const float* negh, const T* qh,
                         const float* qn, float* pd, int* pi, const float* qfloor, int tile0, int tiles, int tps, int QT, int S,
                         int Qpad, int index_base) {
    static std::atomic<uint64_t> lds_set, lds_set_wide;      // per instantiation: devices whose LDS limit is raised
    // FRA_SEARCH_WIDE=0|1 picks the eight-wave or the four-wide-wave kernel (same results; measurements)
    static const bool wide = [] { const char* e = getenv("FRA_SEARCH_WIDE"); return e ? atoi(e) != 0 : SEARCH_WIDE_DEFAULT != 0; }();
    if (wide) {
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&search_kernel_wide<T, KSV>), (int)lds, lds_set_wide)) return rc;
        hipLaunchKernelGGL((search_kernel_wide<T, KSV>), grid, dim3(256), lds, s, G, negh, qh, qn, pd, pi, qfloor, tile0, tiles, tps,
                           QT, S, Qpad, index_base);
        FRA_HIP(hipGetLastError());
        return FRA_OK;
    }
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&search_kernel<T, KSV>), (int)lds, lds_set)) return rc;
    hipLaunchKernelGGL((search_kernel<T, KSV>), grid, dim3(512), lds, s, G, negh, qh, qn, pd, pi, qfloor, tile0, tiles, tps, QT,
                       S, Qpad, index_base);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

template <typename T, int KSV>
static int launch_search_ks(int KS, const dim3 grid, size_t lds, hipStream_t s, const T* G, const float* negh, const T* qh,
                            const float* qn, float* pd, int* pi, const float* qfloor, int tile0, int tiles, int tps, int QT,
                            int S, int Qpad, int index_base) {
    if (KS == KSV) return launch_search<T, KSV>(grid, lds, s, G, negh, qh, qn, pd, pi, qfloor, tile0, tiles, tps, QT, S, Qpad, index_base);
    if constexpr (KSV == 1) return launch_search_ks<T, 5>(KS, grid, lds, s, G, negh, qh, qn, pd, pi, qfloor, tile0, tiles, tps, QT, S, Qpad, index_base);
    else return launch_search<T, 8>(grid, lds, s, G, negh, qh, qn, pd, pi, qfloor, tile0, tiles, tps, QT, S, Qpad, index_base);
}

// Split count for `tiles` gallery tiles and QT query tiles.  A block occupies a compute unit alone (LDS) and all
// blocks of a search do equal work, so the grid should be a whole number of rounds over the chip: among the
// multiples of 8 (the XCD-aware block mapping) up to 64 (=> 256 partial lists per query) take the split count that
// wastes the least of its last round, the smaller one on ties; fewer than 8 tiles get one split per tile.
static void choose_splits(int tiles, int QT, int cus, int* S_out, int* tps_out) {
    int S = tiles < 8 ? tiles : 8;
    if (tiles >= 8) {
        double best = -1.0;
        for (int cand = 8; cand <= 64 && cand <= tiles; cand += 8) {
            const long blocks = (long)QT * cand;
            const long rounds = (blocks + cus - 1) / cus;
            const double eff = (double)blocks / (double)(rounds * cus);
            if (eff > best + 1e-9) { best = eff; S = cand; }
        }
    }
    const int tps = (tiles + S - 1) / S;
    *S_out = (tiles + tps - 1) / tps;  // drop empty splits
    *tps_out = tps;
}

static const int SEARCH_MAX_LISTS = 4 * (64 + 8);   // partial lists per query: main pass + floor pass

#ifndef SEARCH_T1_COEF
#define SEARCH_T1_COEF 0.5
#endif
#ifndef SEARCH_FLOOR_PASS
#define SEARCH_FLOOR_PASS 1
#endif
template <typename T>
static int search_packed(fb_handle* h, const float* queries, int Q, int k, float* dist_out, int32_t* idx_out,
                         hipStream_t s, char* fixed, size_t cap) {
    const int D = h->D;
    const int QT = (Q + BQ - 1) / BQ, Qpad = QT * BQ;
    const int tiles = (int)(h->Npad / BG);
    const int KS = k == 1 ? 1 : (k <= 5 ? 5 : 8);
    const int cus = search_compute_units();
    // A per-lane list starts empty and takes in most of its first few hundred rows, and while ANY of a wave's 512
    // candidates passes its list the wave runs the divergent insertion path; with lists of a few thousand rows
    // (a million-row gallery over 32 splits) that is most chains.  So such a search runs in two passes: the first
    // T1 tiles in one round over the chip, the k-th best distance they give per query as a score floor, then the
    // rest with lists that start at that floor (the fraction of rows that pass it is k / (48 T1) from the first
    // tile on).  T1 balances the first pass's time against the insertions it saves: ~ sqrt(tiles QT / 2), and flat around it.
    int T1 = 0, S1 = 0, tps1 = 0;
    if (SEARCH_FLOOR_PASS) {
        int S0, tps0;
        choose_splits(tiles, QT, cus, &S0, &tps0);
        const double t1 = sqrt(SEARCH_T1_COEF * (double)tiles * (double)QT);
        if (t1 >= 48.0 && t1 <= tiles / 4 && (long)tps0 * (BG / 4) < 150000) {
            S1 = cus / QT < 1 ? 1 : (cus / QT > 8 ? 8 : cus / QT);
            tps1 = ((int)t1 + S1 - 1) / S1;
            T1 = tps1 * S1;
        }
    }
    int S, tps;
    choose_splits(tiles - T1, QT, cus, &S, &tps);
    const int P1 = 4 * S1, P = 4 * S;   // four partial lists (lane groups) per split

    size_t off_qh = 0, off_qn = align_up(off_qh + (size_t)Qpad * D * 2, 256),
           off_fl = align_up(off_qn + (size_t)Qpad * 4, 256),
           off_pd = align_up(off_fl + (size_t)Qpad * 4, 256),
           off_pi = align_up(off_pd + (size_t)(P1 + P) * Qpad * KS * 4, 256),
           total = align_up(off_pi + (size_t)(P1 + P) * Qpad * KS * 4, 256);
    char* sc = nullptr;
    int rc = take_scratch(h, total, fixed, cap, &sc);
    if (rc) return rc;
    // every byte of these arrays that a later kernel reads is written by an earlier kernel of this
    // call (all Qpad query rows, all lists of every query), so stale scratch contents are harmless
    T* qh = reinterpret_cast<T*>(sc + off_qh);
    float* qn = reinterpret_cast<float*>(sc + off_qn);
    float* fl = reinterpret_cast<float*>(sc + off_fl);
    float* pd = reinterpret_cast<float*>(sc + off_pd);
    int* pi = reinterpret_cast<int*>(sc + off_pi);

    hipLaunchKernelGGL(prep_queries_kernel<T>, dim3((Qpad + 3) / 4), dim3(256), 0, s, queries, qh, qn, Q, Qpad, D);
    FRA_HIP(hipGetLastError());
    const size_t lds = 3 * (size_t)SEARCH_BUF_BYTES;
    const T* G = static_cast<const T*>(h->rows);
    if (T1 > 0) {
        rc = launch_search_ks<T, 1>(KS, dim3((unsigned)(QT * S1)), lds, s, G, h->negh, qh, qn, pd, pi, nullptr, 0, T1, tps1, QT, S1,
                                    Qpad, (int)h->index_base);
        if (rc) return rc;
        static_assert(4 * 8 * 8 <= 64 * WSEL, "the first pass's lists (at most 8 splits x 4, 8 entries each) fit one wave");
        hipLaunchKernelGGL(select_topk_wave_kernel, dim3((Qpad + 3) / 4), dim3(256), 0, s, pd, pi, P1, (long)Qpad * KS, (long)KS, KS, k,
                           Qpad, (float*)nullptr, (int*)nullptr, qn, fl);
        FRA_HIP(hipGetLastError());
    }
    const size_t o1 = (size_t)P1 * Qpad * KS;
    rc = launch_search_ks<T, 1>(KS, dim3((unsigned)(QT * S)), lds, s, G, h->negh, qh, qn, pd + o1, pi + o1, T1 > 0 ? fl : nullptr, T1,
                                tiles, tps, QT, S, Qpad, (int)h->index_base);
    if (rc) return rc;
    launch_select(s, Q, pd, pi, P1 + P, (long)Qpad * KS, (long)KS, KS, 0, k, dist_out, idx_out);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

// fp32 path: the Q x N distance matrix is produced and selected from in chunks of queries, so the scratch stays
// within F32_SCRATCH_CAP whatever Q is (one query's row of distances must fit)
static const size_t F32_SCRATCH_CAP = (size_t)256 << 20;
static int search_f32(fb_handle* h, const float* queries, int Q, int k, float* dist_out, int32_t* idx_out,
                      hipStream_t s, char* fixed, size_t cap) {
    FRA_REQUIRE(h->N < 2147483647L, FRA_ERR_ARG, "fb_search: gallery too large");
    const size_t row_bytes = (size_t)h->N * 4;
    const size_t budget = fixed ? cap : F32_SCRATCH_CAP;
    FRA_REQUIRE(row_bytes <= budget, FRA_ERR_ARG,
                "%s: an fp32 gallery of %lld rows needs %zu bytes of scratch per query, more than the %zu available; "
                "store it as f16 / bf16", fixed ? "fb_search_l2_topk_noalloc" : "fb_search", (long long)h->N, row_bytes, budget);
    int qc = (int)(budget / row_bytes);                    // queries per chunk
    if (qc > Q) qc = Q;
    if (fixed)      // the entry point that must not allocate takes what fits in one chunk only
        FRA_REQUIRE(qc == Q, FRA_ERR_ARG,
                    "fb_search_l2_topk_noalloc: this search needs %zu bytes of scratch, the static array holds %zu; "
                    "use fb_search_l2_topk (fewer queries, or a smaller fp32 gallery, fit)", (size_t)Q * row_bytes, cap);
    char* sc = nullptr;
    int rc = take_scratch(h, align_up((size_t)qc * row_bytes, 256), fixed, cap, &sc);
    if (rc) return rc;
    float* dist = reinterpret_cast<float*>(sc);
    for (int q0 = 0; q0 < Q; q0 += qc) {
        const int m = (Q - q0) < qc ? (Q - q0) : qc;
        const long pairs = (long)m * h->N;
        FRA_REQUIRE((pairs + 3) / 4 < 2147483647L, FRA_ERR_ARG, "fb_search: Q*N too large for the fp32 gallery path");
        hipLaunchKernelGGL(l2dist_f32_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, queries + (size_t)q0 * h->D,
                           static_cast<const float*>(h->rows), dist, m, (long)h->N, h->D);
        FRA_HIP(hipGetLastError());
        launch_select(s, m, dist, (const int*)nullptr, 1, 0L, (long)h->N, (int)h->N, (int)h->index_base, k,
                      dist_out + (size_t)q0 * k, idx_out + (size_t)q0 * k);
        FRA_HIP(hipGetLastError());
    }
    return FRA_OK;
}

static int search_any(fb_handle* h, const float* queries, int Q, int k, float* dist_out, int32_t* idx_out,
                      hipStream_t s, char* fixed, size_t cap) {
    if (h->store == FRA_F32) return search_f32(h, queries, Q, k, dist_out, idx_out, s, fixed, cap);
    if (h->store == FRA_F16) return search_packed<f16>(h, queries, Q, k, dist_out, idx_out, s, fixed, cap);
    return search_packed<bf16>(h, queries, Q, k, dist_out, idx_out, s, fixed, cap);
}

// ---- scratch of the entry point that allocates nothing ----------------------------------------------
// One array per GPU the library runs on, held until the process ends.  The name is a host-side stand-in:
// only the address hipGetSymbolAddress returns (with that GPU current) may be handed to kernels.
static const size_t STATIC_SCRATCH_BYTES = (size_t)8 << 20;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
static std::atomic<void*> static_scratch_of[64];   // per device: device address of `scratch`, nullptr = not asked yet

// device address of this GPU's copy of `scratch`; `device` must be the current device
static int static_scratch_ptr(int device, char** out) {
    FRA_REQUIRE(device >= 0 && device < 64, FRA_ERR_ARG, "static scratch: device %d not in [0,64)", device);
    void* p = static_scratch_of[device].load(std::memory_order_acquire);
    if (!p) {
        FRA_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        FRA_REQUIRE(p != nullptr, FRA_ERR_HIP, "static scratch: hipGetSymbolAddress returned a null address");
        static_scratch_of[device].store(p, std::memory_order_release);
    }
    *out = static_cast<char*>(p);
    return FRA_OK;
}

}  // namespace fra

using namespace fra;

extern "C" {

int fb_create(fb_handle** out, int device, const float* gallery, int gallery_is_device, int64_t N, int D,
              int store_dtype, int64_t index_base) {
    FRA_REQUIRE(out != nullptr, FRA_ERR_ARG, "fb_create: null output pointer");
    *out = nullptr;
    FRA_REQUIRE(N > 0 && D > 0, FRA_ERR_ARG, "fb_create: bad shape N=%lld D=%d", (long long)N, D);
    FRA_REQUIRE(store_dtype == FRA_F32 || store_dtype == FRA_F16 || store_dtype == FRA_BF16, FRA_ERR_ARG,
                "fb_create: store_dtype %d unsupported", store_dtype);
    FRA_REQUIRE(store_dtype == FRA_F32 || D == SEARCH_D, FRA_ERR_ARG,
                "fb_create: 16-bit galleries need D == %d (got %d); use FRA_F32 storage", SEARCH_D, D);
    FRA_REQUIRE(index_base >= 0 && index_base + N + 64 < 2147483647LL, FRA_ERR_ARG,
                "fb_create: indices must fit int32 (base %lld + %lld rows)", (long long)index_base, (long long)N);
    int count = 0;
    FRA_HIP(hipGetDeviceCount(&count));
    FRA_REQUIRE(device >= 0 && device < count, FRA_ERR_ARG, "fb_create: device %d not in [0,%d)", device, count);
    FRA_DEVICE(device);
    if (device < 64) {                      // set-up: this GPU's address of the static scratch array
        char* unused = nullptr;
        if (int rc = static_scratch_ptr(device, &unused)) return rc;
    }
    fb_handle* h = new fb_handle();
    h->device = device; h->N = N; h->D = D; h->store = store_dtype; h->index_base = index_base;
    hipError_t e;
    if (store_dtype == FRA_F32) {
        h->Npad = N;
        e = fra::dev_malloc(&h->rows, (size_t)N * D * 4);
    } else {
        h->Npad = (N + BG - 1) / BG * BG;
        e = fra::dev_malloc(&h->rows, (size_t)h->Npad * D * 2);
        if (e == hipSuccess) e = fra::dev_malloc(reinterpret_cast<void**>(&h->negh), (size_t)h->Npad * 4);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess && h->Npad > N) {
            const long total = (h->Npad - N) * (long)D;
            if (store_dtype == FRA_F16)
                hipLaunchKernelGGL(pad_rows_kernel<f16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->copy_stream,
                                   static_cast<f16*>(h->rows), h->negh, (long)N, (long)h->Npad, D);
            else
                hipLaunchKernelGGL(pad_rows_kernel<bf16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->copy_stream,
                                   static_cast<bf16*>(h->rows), h->negh, (long)N, (long)h->Npad, D);
            e = hipStreamSynchronize(h->copy_stream);
        }
    }
    if (e != hipSuccess) {
        set_error("fb_create: allocation failed: %s", hipGetErrorString(e));
        fb_destroy(h);
        return FRA_ERR_HIP;
    }
    if (gallery) {
        int rc = fill_rows(h, 0, gallery, gallery_is_device, N);
        if (rc) { fb_destroy(h); return rc; }
    }
    *out = h;
    return FRA_OK;
}

int fb_fill(fb_handle* h, int64_t row0, const float* rows, int rows_is_device, int64_t n) {
    FRA_REQUIRE(h && rows, FRA_ERR_ARG, "fb_fill: null pointer");
    return fill_rows(h, row0, rows, rows_is_device, n);
}

int fb_fill_async(fb_handle* h, int64_t row0, const float* rows_dev, int64_t n, fra_stream stream) {
    FRA_REQUIRE(h && rows_dev, FRA_ERR_ARG, "fb_fill_async: null pointer");
    hipStream_t s = (hipStream_t)stream;
    return fill_rows(h, row0, rows_dev, 1, n, &s);
}

int64_t fb_rows(const fb_handle* h) { return h ? h->N : -1; }

// packed rows (16-bit [n][D]) and their bias terms, exactly as the search kernel reads them
static int copy_packed(fb_handle* h, int64_t row0, int64_t n, void* rows16, int rows_is_device, float* negh,
                       int negh_is_device, bool to_handle, const char* what) {
    FRA_REQUIRE(h && rows16 && negh, FRA_ERR_ARG, "%s: null pointer", what);
    FRA_REQUIRE(h->store == FRA_F16 || h->store == FRA_BF16, FRA_ERR_ARG, "%s: the gallery stores fp32 rows, nothing is packed",
                what);
    FRA_REQUIRE(row0 >= 0 && n >= 0 && row0 + n <= h->N, FRA_ERR_ARG, "%s: rows [%lld,%lld) outside [0,%lld)", what,
                (long long)row0, (long long)(row0 + n), (long long)h->N);
    if (n == 0) return FRA_OK;
    FRA_DEVICE(h->device);
    char* hr = static_cast<char*>(h->rows) + (size_t)row0 * h->D * 2;
    float* hn = h->negh + row0;
    const size_t rb = (size_t)n * h->D * 2, nb = (size_t)n * 4;
    // on the handle's own stream, synchronised before returning: other streams of the process are not serialised
    if (!h->copy_stream) FRA_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    hipStream_t s = h->copy_stream;
    if (to_handle) {
        FRA_HIP(hipMemcpyAsync(hr, rows16, rb, rows_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
        FRA_HIP(hipMemcpyAsync(hn, negh, nb, negh_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    } else {
        FRA_HIP(hipMemcpyAsync(rows16, hr, rb, rows_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
        FRA_HIP(hipMemcpyAsync(negh, hn, nb, negh_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    }
    FRA_HIP(hipStreamSynchronize(s));
    return FRA_OK;
}

int fb_export_packed(fb_handle* h, int64_t row0, int64_t n, void* rows16_out, int rows_is_device, float* negh_out,
                     int negh_is_device) {
    return copy_packed(h, row0, n, rows16_out, rows_is_device, negh_out, negh_is_device, false, "fb_export_packed");
}

int fb_import_packed(fb_handle* h, int64_t row0, int64_t n, const void* rows16, int rows_is_device, const float* negh,
                     int negh_is_device) {
    return copy_packed(h, row0, n, const_cast<void*>(rows16), rows_is_device, const_cast<float*>(negh), negh_is_device,
                       true, "fb_import_packed");
}

int fb_reserve(fb_handle* h, int max_queries) {
    FRA_REQUIRE(h && max_queries > 0, FRA_ERR_ARG, "fb_reserve: bad argument");
    FRA_DEVICE(h->device);
    if (h->store == FRA_F32) {
        size_t need = (size_t)max_queries * h->N * 4;
        if (need > F32_SCRATCH_CAP) need = F32_SCRATCH_CAP;
        return ensure_scratch(h, align_up(need, 256));
    }
    const int QT = (max_queries + BQ - 1) / BQ, Qpad = QT * BQ;
    const size_t P = SEARCH_MAX_LISTS, KS = 8;   // upper bounds of the partial-list layout in search_packed
    return ensure_scratch(h, align_up((size_t)Qpad * h->D * 2, 256) + 2 * align_up((size_t)Qpad * 4, 256) +
                                 2 * align_up(P * Qpad * KS * 4, 256));
}

int fb_search_l2_topk(fb_handle* h, const float* queries, int Q, int k, float* dist_out, int32_t* idx_out,
                      fra_stream stream) {
    FRA_REQUIRE(h && queries && dist_out && idx_out, FRA_ERR_ARG, "fb_search: null pointer");
    FRA_REQUIRE(Q > 0, FRA_ERR_ARG, "fb_search: Q=%d", Q);
    FRA_REQUIRE(k >= 1 && k <= 8, FRA_ERR_ARG, "fb_search: k=%d not in [1,8]", k);
    FRA_DEVICE(h->device);
    return search_any(h, queries, Q, k, dist_out, idx_out, (hipStream_t)stream, nullptr, 0);
}

int fb_search_l2_topk_noalloc(fb_handle* h, const float* queries, int Q, int k, float* dist_out, int32_t* idx_out,
                              fra_stream stream) {
    FRA_REQUIRE(h && queries && dist_out && idx_out, FRA_ERR_ARG, "fb_search_l2_topk_noalloc: null pointer");
    FRA_REQUIRE(Q > 0, FRA_ERR_ARG, "fb_search_l2_topk_noalloc: Q=%d", Q);
    FRA_REQUIRE(k >= 1 && k <= 8, FRA_ERR_ARG, "fb_search_l2_topk_noalloc: k=%d not in [1,8]", k);
    FRA_DEVICE(h->device);
    char* sc = nullptr;
    if (int rc = static_scratch_ptr(h->device, &sc)) return rc;   // found at fb_create; a lookup, not an allocation
    return search_any(h, queries, Q, k, dist_out, idx_out, (hipStream_t)stream, sc, STATIC_SCRATCH_BYTES);
}

int fb_merge_topk(const float* dist_in, const int32_t* idx_in, int R, int Q, int k, float* dist_out,
                  int32_t* idx_out, fra_stream stream) {
    FRA_REQUIRE(dist_in && idx_in && dist_out && idx_out, FRA_ERR_ARG, "fb_merge: null pointer");
    FRA_REQUIRE(R > 0 && Q > 0 && k >= 1 && k <= 8, FRA_ERR_ARG, "fb_merge: bad shape R=%d Q=%d k=%d", R, Q, k);
    launch_select((hipStream_t)stream, Q, dist_in, idx_in, R, (long)Q * k, (long)k, k, 0, k, dist_out, idx_out);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

int fb_threshold_score(const float* minimum, int64_t* idx_io, float* score_100, int Q, double threshold,
                       fra_stream stream) {
    FRA_REQUIRE(minimum && idx_io && score_100 && Q > 0, FRA_ERR_ARG, "fb_threshold_score: bad argument");
    const float limit = (float)((threshold - 156.0) / (-80.0));  // python float, then compared in fp32
    hipLaunchKernelGGL(threshold_score_kernel, dim3((Q + 255) / 256), dim3(256), 0, (hipStream_t)stream, minimum,
                       reinterpret_cast<long long*>(idx_io), score_100, Q, limit);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

int fb_match_nearest(fb_handle* h, const float* queries, int Q, double threshold, float* minimum, int64_t* idx_out,
                     float* score_100, fra_stream stream) {
    FRA_REQUIRE(h && queries && minimum && idx_out && score_100, FRA_ERR_ARG, "fb_match_nearest: null pointer");
    FRA_REQUIRE(Q > 0, FRA_ERR_ARG, "fb_match_nearest: Q=%d", Q);
    FRA_REQUIRE(h->store == FRA_F32, FRA_ERR_ARG, "fb_match_nearest: the gallery must hold fp32 rows (the demos' exact match)");
    FRA_REQUIRE(h->N >= 1 && h->N < 2147483647L, FRA_ERR_ARG, "fb_match_nearest: gallery of %lld rows", (long long)h->N);
    FRA_DEVICE(h->device);
    hipStream_t s = (hipStream_t)stream;
    const size_t row_bytes = (size_t)h->N * 4;
    FRA_REQUIRE(row_bytes <= F32_SCRATCH_CAP, FRA_ERR_ARG, "fb_match_nearest: gallery too large for the fp32 path");
    int qc = (int)(F32_SCRATCH_CAP / row_bytes);
    if (qc > Q) qc = Q;
    char* sc = nullptr;
    if (int rc = take_scratch(h, align_up((size_t)qc * row_bytes, 256), nullptr, 0, &sc)) return rc;
    float* dist = reinterpret_cast<float*>(sc);
    const float limit = (float)((threshold - 156.0) / (-80.0));  // python float, then compared in fp32
    for (int q0 = 0; q0 < Q; q0 += qc) {
        const int m = (Q - q0) < qc ? (Q - q0) : qc;
        const long pairs = (long)m * h->N;
        FRA_REQUIRE((pairs + 3) / 4 < 2147483647L, FRA_ERR_ARG, "fb_match_nearest: Q*N too large");
        hipLaunchKernelGGL(l2dist_f32_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, queries + (size_t)q0 * h->D,
                           static_cast<const float*>(h->rows), dist, m, (long)h->N, h->D);
        hipLaunchKernelGGL(match_nearest_kernel, dim3(m), dim3(256), 0, s, dist, (long)h->N, (int)h->index_base, limit,
                           minimum + q0, reinterpret_cast<long long*>(idx_out) + q0, score_100 + q0);
        FRA_HIP(hipGetLastError());
    }
    return FRA_OK;
}

int fb_destroy(fb_handle* h) {
    if (!h) return FRA_OK;
    fra::DeviceGuard guard(h->device);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->rows) (void)fra::dev_free(h->rows);
    if (h->negh) (void)fra::dev_free(h->negh);
    if (h->scratch) (void)fra::dev_free(h->scratch);
    for (int i = 0; i < h->n_retired; ++i) (void)fra::dev_free(h->retired[i]);
    delete h;
    return FRA_OK;
}

}  // extern "C"
