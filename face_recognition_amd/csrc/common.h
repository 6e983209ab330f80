// Shared declarations for the gfx950 kernels and the C-ABI layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "../../include/face_recognition_amd.h"  // status codes and dtype tags (global namespace)

namespace fra {

typedef _Float16 f16;
typedef __bf16 bf16;

template <typename T> struct vec8 { typedef T type __attribute__((ext_vector_type(8))); };
template <typename T> struct vec4 { typedef T type __attribute__((ext_vector_type(4))); };
template <typename T> struct vec16 { typedef T type __attribute__((ext_vector_type(16))); };
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// activation / weight element types selectable at handle creation
enum ActDtype { ACT_F32 = 0, ACT_BF16 = 1, ACT_F16 = 2 };

// one 32x32x16 MFMA step, fp32 accumulate.  Operand maps (lane l, r = l & 31, h = l >> 5):
//   a[j] = A[row r][k = 8h + j],  b[j] = B[k = 8h + j][col r],
//   acc[g] = D[row (g & 3) + 8 (g >> 2) + 4 h][col r]
__device__ __forceinline__ f32x16 mfma32(vec8<f16>::type a, vec8<f16>::type b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(vec8<bf16>::type a, vec8<bf16>::type b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// row of the 32x32 accumulator tile held in register g by a lane of half h
__device__ __forceinline__ f32x4 mfma16(vec8<f16>::type a, vec8<f16>::type b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(vec8<bf16>::type a, vec8<bf16>::type b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// one sparse 16x16x64 MFMA step (A 2:4 structured: eight values per lane, their positions in `idx`), fp32 accumulate
__device__ __forceinline__ f32x4 smfma16(vec8<f16>::type a, vec16<f16>::type b, f32x4 c, int idx) {
    return __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, idx, 0, 0);
}
__device__ __forceinline__ f32x4 smfma16(vec8<bf16>::type a, vec16<bf16>::type b, f32x4 c, int idx) {
    return __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, c, idx, 0, 0);
}
// min(v, 0) as ONE instruction: the median of (v, 0, a huge negative number).  fminf() costs two (IEEE minimum
// first quietens a signalling NaN with a v_max v, v); activations here are finite and far above -3e38.
__device__ __forceinline__ float neg_part(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, -3.0e38f); }
__device__ __forceinline__ int acc_row(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
const char* last_error();

// Status codes come from the public header: FRA_ERR_ARG (bad argument), FRA_ERR_STATE (call
// order), FRA_ERR_HIP (a HIP runtime call failed), FRA_ERR_WEIGHTS (missing / mis-shaped tensor),
// FRA_ERR_WORKSPACE (workspace too small).

#define FRA_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            fra::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,   \
                           __LINE__);                                                          \
            return FRA_ERR_HIP;                                                           \
        }                                                                                      \
    } while (0)

#define FRA_REQUIRE(cond, code, ...)                                                           \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            fra::set_error(__VA_ARGS__);                                                       \
            return code;                                                                       \
        }                                                                                      \
    } while (0)

// Makes `device` current for the lifetime of the object and restores the caller's device afterwards: an entry point
// must not leave the thread on another device than it found it on (PyTorch keeps its own idea of the current one).
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
        else if (err == hipSuccess) prev = -1;             // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define FRA_DEVICE(device)                 \
    fra::DeviceGuard guard__(device);      \
    FRA_HIP(guard__.err)

// A kernel's dynamic-LDS limit is a property of the kernel ON ONE DEVICE: raise it the first time the kernel
// is launched on each device of the process (bit d of `done` = device d has it; devices past 63 set it on
// every launch).  *device, if given, receives the current device.
inline int ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done, int* device = nullptr) {
    int dev = 0;
    FRA_HIP(hipGetDevice(&dev));
    const bool known = dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1);
    if (!known) {
        FRA_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        if (dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    }
    if (device) *device = dev;
    return FRA_OK;
}

// ---- device memory: every allocation and release of the library goes through these two, which count their calls
// (fra_dbg_alloc_calls): the entry point that promises to allocate nothing is tested by that count standing still
hipError_t dev_malloc(void** p, size_t bytes);
hipError_t dev_free(void* p);
template <typename T> inline hipError_t dev_malloc(T** p, size_t bytes) { return dev_malloc(reinterpret_cast<void**>(p), bytes); }
long alloc_calls();

// ---- waits that give up ---------------------------------------------------------------------------
// The kernels that synchronise their waves through words in LDS poll with a bound, so that a protocol bug cannot hang
// the GPU.  A poll that runs out stores a tag (two letters naming the kernel << 16 | wave) to a host-visible word of the
// device it ran on and goes on with whatever LDS holds: the launch's results are wrong, and the next entry point that
// checks the word reports it instead of returning them with FRA_OK.
unsigned* fault_word();                 // device pointer to the current device's word (nullptr: it could not be allocated)
int spin_limit(int dflt);               // polls before a wait gives up: fra_dbg_set_spin_limit()'s value, else dflt
int check_device_fault(int device);     // FRA_ERR_STATE + message if a wait gave up on `device` since the last check; clears the word

// ---- kernel launchers (device pointers, parameters already in kernel layout) ------------------
// All return a Status; all launch on `s` and never synchronise.

// stem: x f32 NCHW [B,3,112,112] -> out T NHWC [B,56,56,64]; w f32 [27][64] (tap-major), bias/slope f32 [64]
int launch_stem(int dt, const float* x, const float* w, const float* bias, const float* slope, void* out, int B,
                hipStream_t s);
// depthwise 3x3, pad 1: in T [B,H,W,C] -> out T [B,Ho,Wo,C]; w T [9][C]; slope == nullptr -> linear
int launch_dw3x3(int dt, const void* in, const void* w, const float* bias, const float* slope, void* out, int B,
                 int H, int W, int C, int stride, hipStream_t s);
// pointwise conv as GEMM: out[M,N] = act(in[M,K] . w[N,K]^T + bias) (+ resid[M,N])
int launch_pw(int dt, const void* in, const void* w, const float* bias, const float* slope, const void* resid,
              void* out, int64_t M, int K, int N, hipStream_t s);
// 7x7 global depthwise conv: in T [B,49,C] -> out T [B,C]; w T [49][C]
int launch_gdconv7(int dt, const void* in, const void* w, const float* bias, void* out, int B, int C, hipStream_t s);
// fused tail: 1x1 conv 128 -> 512 + PReLU on the 7x7 map, then the 7x7 global depthwise conv (linear).
// x T [B][49][128]; w6 T [512][128]; b6, s6 f32 [512] (s6 = PReLU slope MINUS ONE); w7 T [49][512]; b7 f32 [512]; out T [B][512]
int launch_tail(int dt, const void* x, const void* w6, const float* b6, const float* s6, const void* w7, const float* b7,
                void* out, int B, hipStream_t s);
// embedding head: out f32 [B,E] = l2norm(in[B,K] . w[E,K]^T + bias)
int launch_linear_l2norm(int dt, const void* in, const void* w, const float* bias, float* out, int B, int K, int E,
                         hipStream_t s);
// out = l2norm(a + b) row-wise (b may be nullptr: plain l2norm), f32 [B,E]
int launch_add_l2norm(const float* a, const float* b, float* out, int B, int E, hipStream_t s);
// fused inverted-residual block (expand 1x1 + PReLU, depthwise 3x3 + PReLU, project 1x1, optional shortcut).
// "p" arrays are indexed by expanded channel in the permuted order perm_pos() defines.
struct BlockParams {
    int cin, cexp, cout, h, stride, residual;
    const void* w1;     // T [cexp][cin], rows in natural channel order, BN folded
    const float* b1p;   // [cexp] permuted
    const float* s1p;   // [cexp] permuted, PReLU slope MINUS ONE
    const void* w2p;    // T [9][cexp] permuted, tap-major
    const float* b2p;
    const float* s2p;   // [cexp] permuted, PReLU slope MINUS ONE
    const void* w3p;    // T, cout * cexp elements in the order w3_image_index() defines (the project stage's LDS image)
    const float* b3;    // [cout]
};
// position of natural expanded channel c = 16g + 8a + 4h + i in the permuted order: 16g + 8h + 4a + i
inline int perm_pos(int c) { return (c & ~15) | ((c & 4) << 1) | ((c & 8) >> 1) | (c & 3); }
// Where element (output channel co, expanded-channel POSITION pos = perm_pos(channel)) of the project weights is
// stored: slice-major (64 positions per slice), inside a slice exactly the order the kernel keeps in LDS, so that
// a slice is copied with 16-byte LDS-DMA pieces.  Row 32 n + 16 (g & 1) + m of a slice holds, for output channel
// 16 n + m, four 16-byte units [pair P][g >> 1]; unit element j is position 32 P + 4 g + j (j < 4) or
// 32 P + 16 + 4 g + (j - 4): k-group g of the project MFMA whose B operand a lane builds from two 16-channel chunks.
inline size_t w3_image_index(int co, int pos, int cout) {
    const int c = pos >> 6, P = (pos >> 5) & 1, hi = (pos >> 4) & 1, g = (pos >> 2) & 3, i = pos & 3;
    const int row = 32 * (co >> 4) + 16 * (g & 1) + (co & 15), unit = 2 * P + (g >> 1), j = 4 * hi + i;
    return (((size_t)c * 2 * cout + row) * 4 + unit) * 8 + j;
}
int launch_block(int dt, const BlockParams& p, const void* x, void* out, int B, hipStream_t s);
// the same block as a row pipeline without workgroup-wide barriers (kernels_pblock.hip): same results bit for bit;
// pblock_has(): the shapes it is built for; pblock_pays(): whether it is the faster of the two for a batch of B images
bool pblock_has(const BlockParams& p);
bool pblock_pays(const BlockParams& p, int B);
// what launch_block() does with a batch of B images: true = the row pipeline
bool block_uses_pipeline(const BlockParams& p, int B);
int launch_pblock(int dt, const BlockParams& p, const void* x, void* out, int B, hipStream_t s);

// ---- residual blocks with the activation kept in registers between blocks (kernels_rblock.hip) ----------------
// Everything one 64-channel slice of one block needs travels as ONE package, stored in global memory exactly as it
// sits in LDS (padding included), so a package is copied by 1 KB LDS-DMA pieces:
//   expand group: W1 image | b1 | s1 (rounded up to whole pieces)
//   depthwise / project group: W3 image | b2 | s2 | 3x3 weight image | b3      (rounded up to whole pieces)
// The expand stage reads the first group only, the depthwise + project stage and the block epilogue the second only, so
// a kernel may keep the two in separate LDS buffers and load them at different times.
// W1 image: 4 channel tiles x 32 rows of C/2 elements + one 16-byte unit of padding; row 16 p + m of tile t holds,
//   for expanded channel 64 c + 16 t + m, the units [ks][g >> 1] of the k-groups g with g & 1 == p, where element j
//   of k-step ks, k-group g is INPUT channel rb_in_channel(ks, g, j) -- the order in which a lane of the project
//   stage holds its output channels, so that a block's output is the next block's expand operand as it stands.
// b1, s1 (slope - 1): natural channel order inside the slice.  Position of expanded channel 16 t + 4 g + i of the
// slice on the depthwise / project side: 16 g + 4 t + i (rb_pos); b2, s2, the 3x3 weight image (dw_wimage_bytes(64))
// and the W3 image's columns use positions.
// W3 image: 2 C rows of 64 + 16 bytes; row 32 n + 16 (g & 1) + m holds, for output channel 16 n + m, units
// [pair P][g >> 1]; element j of a unit is position 32 P + 4 g + j (j < 4) or 32 P + 16 + 4 g + (j - 4).
// ---- the 3x3 weight image of the fused block kernels in LDS -------------------------------------------------
// One row per window row p of two planes: plane 0 holds, per expanded-channel position, the 32-bit word
// tap (p,0) | tap (p,1) << 16, plane 1 the word tap (p,2) (upper half zero).  A plane is `ch` words + 16, so
// that the two planes a wave's 32-bit read touches sit on different banks.
constexpr int dw_wplane(int ch) { return ch * 4 + 64; }
constexpr int dw_wrow(int ch) { return 2 * dw_wplane(ch); }
constexpr int dw_wimage_bytes(int ch) { return 3 * dw_wrow(ch); }
template <int C> struct RbPkg {
    static constexpr int PW1 = C + 16, W1_BYTES = 128 * PW1;
    static constexpr int OFF_B1 = W1_BYTES, OFF_S1 = OFF_B1 + 256;
    static constexpr int EG_BYTES = (OFF_S1 + 256 + 1023) / 1024 * 1024, EG_PIECES = EG_BYTES / 1024;   // expand group
    static constexpr int PW3 = 80, W3_BYTES = 2 * C * PW3, OFF_W3 = EG_BYTES;
    static constexpr int OFF_B2 = OFF_W3 + W3_BYTES, OFF_S2 = OFF_B2 + 256;
    static constexpr int OFF_W2 = OFF_S2 + 256, OFF_B3 = OFF_W2 + dw_wimage_bytes(64);
    static constexpr int RAW = OFF_B3 + C * 4;
    static constexpr int BYTES = (RAW + 1023) / 1024 * 1024, PIECES = BYTES / 1024;
    static constexpr int DG_BYTES = BYTES - EG_BYTES, DG_PIECES = DG_BYTES / 1024;                      // depthwise / project group
};
inline int rb_in_channel(int ks, int g, int j) { return 16 * (2 * ks + (j >> 2)) + 4 * g + (j & 3); }
inline int rb_pos(int e_local) { return 16 * ((e_local >> 2) & 3) + 4 * (e_local >> 4) + (e_local & 3); }
// bytes of the packages of one block (cexp / 64 slices); 0 for an unsupported width
size_t rblock_package_bytes(int c, int cexp);
// folded weights in natural order (w1 [cexp][c], w2 [cexp][9], w3 [c][cexp]; s1 / s2 = PReLU slopes) -> packages
int pack_rblock(int dt, int c, int cexp, const float* w1, const float* b1, const float* s1, const float* w2,
                const float* b2, const float* s2, const float* w3, const float* b3, void* out);
// `nblk` consecutive blocks (packages back to back at wp) on x -> out, T NHWC [B][h][h][c]; nblk > 1 keeps the
// activation in registers between blocks (whole-image workgroups); out may be x
int launch_rblock(int dt, int c, int cexp, int h, const void* x, const void* wp, int nblk, void* out, int B, hipStream_t s);
// the same chain as four wide waves per workgroup, one per SIMD (kernels_wblock.hip): same packages, same results bit
// for bit; wblock_has(): the shapes it is built for
// 0: every batch size runs the same arithmetic per output value (a face's embedding does not depend on the batch it is
// computed in, bit for bit); 1 (default): batches whose band launches fit the chip in one go (7 B <= compute units) add a
// block's project sum in two halves (kernels_rblock.hip, dual slices) -- shorter launches, the same values to rounding
int small_batch_mode();
void set_small_batch_mode(int mode);
bool wblock_has(int c, int cexp, int h);
int launch_wblock(int dt, int c, int cexp, int h, const void* x, const void* wp, int nblk, void* out, int B, hipStream_t s);
// true when `nblk` > 1 blocks of this shape should run as one launch for a batch of B images
bool rblock_chain_pays(int c, int cexp, int h, int B);
// blocks the next launch covers when `remaining` identical blocks are left: all of them (whole-image workgroups), two
// (bands of a small batch, each computing its neighbours' nearest rows itself), or one
int rblock_chain_blocks(int c, int cexp, int h, int B, int remaining);
// fused head: stem conv (3->64, 3x3 s2, PReLU) + depthwise 3x3 (64 ch, PReLU) -> out T NHWC [B,56,56,64].
// x: n_src crops, f32 NCHW in [-1,1] or (x_is_u8) uint8 HWC; images n_src..B-1 are mirrored copies of images 0..B-n_src-1
struct StemDwParams {
    const void* wa;     // T [64][32]: row 16 (2 (U >> 2) + a) + 4 (U & 3) + i holds output channel 8 U + 4 a + i; columns = taps (c*9+ky*3+kx), 27..31 zero
    const float* b1;    // [64] natural order
    const float* s1;    // [64] PReLU slope MINUS ONE
    const void* w2;     // T [9][64] tap-major depthwise weights
    const float* b2;
    const float* s2;    // [64] PReLU slope MINUS ONE
};
int launch_stemdw(int dt, const StemDwParams& p, const void* x, int x_is_u8, int n_src, void* out, int B, hipStream_t s);
// T NHWC [n] -> f32 (debug taps)
int launch_to_f32(int dt, const void* in, float* out, int64_t n, hipStream_t s);

}  // namespace fra
