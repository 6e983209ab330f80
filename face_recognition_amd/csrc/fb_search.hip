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

#include "common.h"

struct fb_handle {
    int device = 0;
    int64_t N = 0;      // rows
    int64_t Npad = 0;   // rows allocated (multiple of 64 for the packed layout)
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
};

namespace fra {

static const int SEARCH_D = 512;   // feature width of the MFMA search kernel
static const int BQ = 256;         // queries per block (8 waves x 32)
static const int BG = 64;          // gallery rows per LDS tile

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
// MFMA search.  Block = 8 waves; wave w owns queries [q0 + 32 w, +32) whose fragments stay in
// registers for the whole kernel (32 k-steps x 4 VGPRs).  Gallery tiles of 64 rows (64 KB, plus
// their 64 bias terms) arrive in LDS by direct global->LDS loads, double buffered; the 16-byte
// chunk c of row r sits at chunk position c ^ (r & 15), applied on the global source address
// because the LDS side of such a load is lane-linear.
// Each lane keeps KS (score, row) pairs, best first, for its query and its half of the rows.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* GPTR;  // global source of an LDS-DMA load
typedef __attribute__((address_space(3))) void* LPTR;        // its (wave-uniform) LDS destination

template <int KS> struct TopList {
    float s[KS];
    int id[KS];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < KS; ++j) { s[j] = -INFINITY; id[j] = -1; }
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

template <typename T, int KS>
__global__ __launch_bounds__(512) void search_kernel(const T* __restrict__ G, const float* __restrict__ negh,
                                                     const T* __restrict__ Qh, const float* __restrict__ qn,
                                                     float* __restrict__ pdist, int* __restrict__ pidx,
                                                     int tiles_total, int tiles_per_split, int QT, int S, int Qpad,
                                                     int index_base) {
    typedef typename vec8<T>::type V8;
    constexpr int D = SEARCH_D, CH = D / 8, KSTEPS = D / 16;
    constexpr int TILE_BYTES = BG * D * 2;          // 65536
    constexpr int BUF_BYTES = TILE_BYTES + BG * 4;  // + bias terms
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // block -> (query tile, gallery split); blocks that share blockIdx % 8 share an XCD's L2, so
    // let them walk the same gallery range with different query tiles
    int qt, split;
    {
        const int b = blockIdx.x;
        if ((S & 7) == 0) { const int j = b >> 3; split = (b & 7) + 8 * (j / QT); qt = j % QT; }
        else { qt = b % QT; split = b / QT; }
    }
    const int t_begin = split * tiles_per_split;
    int t_end = t_begin + tiles_per_split;
    if (t_end > tiles_total) t_end = tiles_total;

    // query fragments (MFMA B operand): lane (r, h) holds Q[q][16 s + 8 h .. +8]
    const int qrow = qt * BQ + wave * 32 + r;
    V8 qf[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) qf[s] = *reinterpret_cast<const V8*>(Qh + (size_t)qrow * D + s * 16 + h * 8);

    TopList<KS> top;
    top.init();

    // stage one tile: 64 rows x 64 chunks; wave w stages rows w, w+8, ...; one row per instruction
    auto stage = [&](int tile, int bufsel) {
        char* base = smem + bufsel * BUF_BYTES;
        const T* gt = G + (size_t)tile * BG * D;
#pragma unroll
        for (int i = 0; i < BG / 8; ++i) {
            const int row = i * 8 + wave;
            const int c = lane ^ (row & 15);
            __builtin_amdgcn_global_load_lds((GPTR)(gt + (size_t)row * D + c * 8), (LPTR)(base + row * (CH * 16)), 16, 0,
                                             0);
        }
        if (wave == 0)
            __builtin_amdgcn_global_load_lds((GPTR)(negh + (size_t)tile * BG + lane), (LPTR)(base + TILE_BYTES), 4, 0, 0);
    };

    if (t_begin < t_end) stage(t_begin, 0);
    __syncthreads();

    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        if (t + 1 < t_end) stage(t + 1, cur ^ 1);
        const char* base = smem + cur * BUF_BYTES;
        const float* nh = reinterpret_cast<const float*>(base + TILE_BYTES);

        f32x16 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(nh + mt * 32 + 8 * q + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[mt][4 * q + i] = b4[i];
            }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int c = 2 * s + h;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = mt * 32 + r;
                const V8 a = *reinterpret_cast<const V8*>(base + ((row * CH + (c ^ (row & 15))) << 4));
                acc[mt] = mfma32(a, qf[s], acc[mt]);
            }
        }
        // top-k update: cheap reject first (max of the 32 new scores against the lane's k-th best)
        float mx = acc[0][0];
#pragma unroll
        for (int g = 1; g < 16; ++g) mx = fmaxf(mx, acc[0][g]);
#pragma unroll
        for (int g = 0; g < 16; ++g) mx = fmaxf(mx, acc[1][g]);
        if (__any(mx > top.s[KS - 1])) {
            const int row0 = t * BG;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float v = acc[mt][g];
                    if (__any(v > top.s[KS - 1])) {
                        if (v > top.s[KS - 1]) top.insert(v, row0 + mt * 32 + acc_row(g, h));
                    }
                }
        }
        __syncthreads();  // next tile landed (the compiler drains the LDS-DMA here); this buffer is free
    }

    // partial lists: [P = 2 S][Qpad][KS], list p = 2 split + h
    const float q2 = qn[qrow];
    const size_t o = ((size_t)(2 * split + h) * Qpad + qrow) * KS;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        const bool ok = top.id[j] >= 0;
        pdist[o + j] = ok ? fmaxf(q2 - 2.0f * top.s[j], 0.0f) : INFINITY;
        pidx[o + j] = ok ? top.id[j] + index_base : -1;
    }
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
    for (int round = 0; round < k; ++round) {
        float bd = INFINITY;
        int bi = 0x7fffffff;
        for (long c = tid; c < total; c += 256) {
            const long l = c / len, j = c - l * len;
            const long a = l * strideL + (long)q * strideQ + j;
            const float d = dist[a];
            const int i = idx ? idx[a] : (int)j + index_base;
            if (i < 0) continue;
            if (round > 0 && !pair_less(last_d, last_i, d, i)) continue;
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

__global__ __launch_bounds__(256) void threshold_score_kernel(const float* __restrict__ minimum,
                                                              long long* __restrict__ idx,
                                                              float* __restrict__ score, int Q, float limit) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Q) return;
    const float m = minimum[i];
    if (m > limit) idx[i] = -1;                                   // cam_demo.py:100
    // cam_demo.py:105 -- multiply and add round separately in the reference, so no fused multiply-add
    const float sc = __fadd_rn(__fmul_rn(m, -80.0f), 156.0f);
    score[i] = fminf(fmaxf(sc, 0.0f), 100.0f);
}

// ------------------------------------------------------------------------------------------------
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int ensure_scratch(fb_handle* h, size_t bytes) {
    if (h->scratch_bytes >= bytes) return FRA_OK;
    size_t want = bytes > 2 * h->scratch_bytes ? bytes : 2 * h->scratch_bytes;   // geometric growth bounds the retired list
    void* fresh = nullptr;
    FRA_HIP(hipMalloc(&fresh, want));
    if (h->scratch) {
        if (h->n_retired < 32) {
            h->retired[h->n_retired++] = h->scratch;
        } else {                       // cannot happen with doubling (2^32 growth); drain and free rather than leak
            (void)hipDeviceSynchronize();
            (void)hipFree(h->scratch);
        }
    }
    h->scratch = fresh;
    h->scratch_bytes = want;
    return FRA_OK;
}

static int fill_rows(fb_handle* h, int64_t row0, const float* src, int is_device, int64_t n) {
    FRA_REQUIRE(row0 >= 0 && n >= 0 && row0 + n <= h->N, FRA_ERR_ARG, "fb_fill: rows [%lld,%lld) outside [0,%lld)",
                (long long)row0, (long long)(row0 + n), (long long)h->N);
    if (n == 0) return FRA_OK;
    FRA_HIP(hipSetDevice(h->device));
    const int D = h->D;
    if (h->store == FRA_F32) {
        FRA_HIP(hipMemcpy(static_cast<float*>(h->rows) + row0 * D, src, (size_t)n * D * 4,
                          is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
        return FRA_OK;
    }
    const int64_t step = is_device ? n : (int64_t)(64 << 20) / (D * 4);  // host data goes through a 64 MB bounce
    float* bounce = nullptr;
    if (!is_device) FRA_HIP(hipMalloc(&bounce, (size_t)step * D * 4));
    int rc = FRA_OK;
    for (int64_t s = 0; s < n && rc == FRA_OK; s += step) {
        const int64_t m = (n - s) < step ? (n - s) : step;
        const float* dsrc = src + s * D;
        if (!is_device) {
            hipError_t e = hipMemcpy(bounce, src + s * D, (size_t)m * D * 4, hipMemcpyHostToDevice);
            if (e != hipSuccess) { set_error("fb_fill: hipMemcpy failed: %s", hipGetErrorString(e)); rc = FRA_ERR_HIP; break; }
            dsrc = bounce;
        }
        const unsigned blocks = (unsigned)((m + 3) / 4);
        if (h->store == FRA_F16)
            hipLaunchKernelGGL(pack_rows_kernel<f16>, dim3(blocks), dim3(256), 0, 0, dsrc,
                               static_cast<f16*>(h->rows) + (row0 + s) * D, h->negh + row0 + s, (long)m, D);
        else
            hipLaunchKernelGGL(pack_rows_kernel<bf16>, dim3(blocks), dim3(256), 0, 0, dsrc,
                               static_cast<bf16*>(h->rows) + (row0 + s) * D, h->negh + row0 + s, (long)m, D);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { set_error("fb_fill: pack kernel failed: %s", hipGetErrorString(e)); rc = FRA_ERR_HIP; }
    }
    if (bounce) (void)hipFree(bounce);
    return rc;
}

template <typename T>
static int search_packed(fb_handle* h, const float* queries, int Q, int k, float* dist_out, int32_t* idx_out,
                         hipStream_t s) {
    const int D = h->D;
    const int QT = (Q + BQ - 1) / BQ, Qpad = QT * BQ;
    const int tiles = (int)(h->Npad / BG);
    const int KS = k == 1 ? 1 : (k <= 5 ? 5 : 8);
    // splits: enough blocks to fill the chip a few times over, at most 64 (=> 128 lists per query)
    int S = (1024 + QT - 1) / QT;
    if (S > 64) S = 64;
    if (S > tiles) S = tiles;
    if (S >= 8) S &= ~7;  // multiple of 8 enables the XCD-aware block mapping
    const int tps = (tiles + S - 1) / S;
    S = (tiles + tps - 1) / tps;  // drop empty splits
    const int P = 2 * S;

    size_t off_qh = 0, off_qn = align_up(off_qh + (size_t)Qpad * D * 2, 256),
           off_pd = align_up(off_qn + (size_t)Qpad * 4, 256),
           off_pi = align_up(off_pd + (size_t)P * Qpad * KS * 4, 256),
           total = align_up(off_pi + (size_t)P * Qpad * KS * 4, 256);
    int rc = ensure_scratch(h, total);
    if (rc) return rc;
    char* sc = static_cast<char*>(h->scratch);
    T* qh = reinterpret_cast<T*>(sc + off_qh);
    float* qn = reinterpret_cast<float*>(sc + off_qn);
    float* pd = reinterpret_cast<float*>(sc + off_pd);
    int* pi = reinterpret_cast<int*>(sc + off_pi);

    hipLaunchKernelGGL(prep_queries_kernel<T>, dim3((Qpad + 3) / 4), dim3(256), 0, s, queries, qh, qn, Q, Qpad, D);
    FRA_HIP(hipGetLastError());
    const size_t lds = 2 * (size_t)(BG * SEARCH_D * 2 + BG * 4);
    const dim3 grid((unsigned)(QT * S)), block(512);
    const T* G = static_cast<const T*>(h->rows);
#define FRA_LAUNCH_SEARCH(KSV)                                                                                   \
    do {                                                                                                         \
        FRA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&search_kernel<T, KSV>),                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
        hipLaunchKernelGGL((search_kernel<T, KSV>), grid, block, lds, s, G, h->negh, qh, qn, pd, pi, tiles, tps, \
                           QT, S, Qpad, (int)h->index_base);                                                     \
    } while (0)
    if (KS == 1) FRA_LAUNCH_SEARCH(1);
    else if (KS == 5) FRA_LAUNCH_SEARCH(5);
    else FRA_LAUNCH_SEARCH(8);
#undef FRA_LAUNCH_SEARCH
    FRA_HIP(hipGetLastError());
    hipLaunchKernelGGL(select_topk_kernel, dim3(Q), dim3(256), 0, s, pd, pi, P, (long)Qpad * KS, (long)KS, KS, 0, k,
                       dist_out, idx_out);
    FRA_HIP(hipGetLastError());
    return FRA_OK;
}

static int search_f32(fb_handle* h, const float* queries, int Q, int k, float* dist_out, int32_t* idx_out,
                      hipStream_t s) {
    const size_t bytes = (size_t)Q * h->N * 4;
    int rc = ensure_scratch(h, align_up(bytes, 256));
    if (rc) return rc;
    float* dist = static_cast<float*>(h->scratch);
    const long pairs = (long)Q * h->N;
    FRA_REQUIRE((pairs + 3) / 4 < 2147483647L, FRA_ERR_ARG, "fb_search: Q*N too large for the fp32 gallery path");
    hipLaunchKernelGGL(l2dist_f32_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, queries,
                       static_cast<const float*>(h->rows), dist, Q, (long)h->N, h->D);
    FRA_HIP(hipGetLastError());
    FRA_REQUIRE(h->N < 2147483647L, FRA_ERR_ARG, "fb_search: gallery too large");
    hipLaunchKernelGGL(select_topk_kernel, dim3(Q), dim3(256), 0, s, dist, (const int*)nullptr, 1, 0L, (long)h->N,
                       (int)h->N, (int)h->index_base, k, dist_out, idx_out);
    FRA_HIP(hipGetLastError());
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
    FRA_HIP(hipSetDevice(device));
    fb_handle* h = new fb_handle();
    h->device = device; h->N = N; h->D = D; h->store = store_dtype; h->index_base = index_base;
    hipError_t e;
    if (store_dtype == FRA_F32) {
        h->Npad = N;
        e = hipMalloc(&h->rows, (size_t)N * D * 4);
    } else {
        h->Npad = (N + BG - 1) / BG * BG;
        e = hipMalloc(&h->rows, (size_t)h->Npad * D * 2);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->negh), (size_t)h->Npad * 4);
        if (e == hipSuccess && h->Npad > N) {
            const long total = (h->Npad - N) * (long)D;
            if (store_dtype == FRA_F16)
                hipLaunchKernelGGL(pad_rows_kernel<f16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0,
                                   static_cast<f16*>(h->rows), h->negh, (long)N, (long)h->Npad, D);
            else
                hipLaunchKernelGGL(pad_rows_kernel<bf16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0,
                                   static_cast<bf16*>(h->rows), h->negh, (long)N, (long)h->Npad, D);
            e = hipDeviceSynchronize();
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

int64_t fb_rows(const fb_handle* h) { return h ? h->N : -1; }

int fb_reserve(fb_handle* h, int max_queries) {
    FRA_REQUIRE(h && max_queries > 0, FRA_ERR_ARG, "fb_reserve: bad argument");
    FRA_HIP(hipSetDevice(h->device));
    if (h->store == FRA_F32) return ensure_scratch(h, align_up((size_t)max_queries * h->N * 4, 256));
    const int QT = (max_queries + BQ - 1) / BQ, Qpad = QT * BQ;
    const size_t P = 128, KS = 8;   // upper bounds of the partial-list layout in search_packed
    return ensure_scratch(h, align_up((size_t)Qpad * h->D * 2, 256) + align_up((size_t)Qpad * 4, 256) +
                                 2 * align_up(P * Qpad * KS * 4, 256));
}

int fb_search_l2_topk(fb_handle* h, const float* queries, int Q, int k, float* dist_out, int32_t* idx_out,
                      fra_stream stream) {
    FRA_REQUIRE(h && queries && dist_out && idx_out, FRA_ERR_ARG, "fb_search: null pointer");
    FRA_REQUIRE(Q > 0, FRA_ERR_ARG, "fb_search: Q=%d", Q);
    FRA_REQUIRE(k >= 1 && k <= 8, FRA_ERR_ARG, "fb_search: k=%d not in [1,8]", k);
    FRA_HIP(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (h->store == FRA_F32) return search_f32(h, queries, Q, k, dist_out, idx_out, s);
    if (h->store == FRA_F16) return search_packed<f16>(h, queries, Q, k, dist_out, idx_out, s);
    return search_packed<bf16>(h, queries, Q, k, dist_out, idx_out, s);
}

int fb_merge_topk(const float* dist_in, const int32_t* idx_in, int R, int Q, int k, float* dist_out,
                  int32_t* idx_out, fra_stream stream) {
    FRA_REQUIRE(dist_in && idx_in && dist_out && idx_out, FRA_ERR_ARG, "fb_merge: null pointer");
    FRA_REQUIRE(R > 0 && Q > 0 && k >= 1 && k <= 8, FRA_ERR_ARG, "fb_merge: bad shape R=%d Q=%d k=%d", R, Q, k);
    hipLaunchKernelGGL(select_topk_kernel, dim3(Q), dim3(256), 0, (hipStream_t)stream, dist_in, idx_in, R,
                       (long)Q * k, (long)k, k, 0, k, dist_out, idx_out);
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

int fb_destroy(fb_handle* h) {
    if (!h) return FRA_OK;
    (void)hipSetDevice(h->device);
    if (h->rows) (void)hipFree(h->rows);
    if (h->negh) (void)hipFree(h->negh);
    if (h->scratch) (void)hipFree(h->scratch);
    for (int i = 0; i < h->n_retired; ++i) (void)hipFree(h->retired[i]);
    delete h;
    return FRA_OK;
}

}  // extern "C"
