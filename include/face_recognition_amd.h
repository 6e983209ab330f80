/*
 * face_recognition_amd -- C ABI of the MI355X (gfx950) face-embedding and facebank-search library.
 *
 * The reference project (liusheng134/Face_Recognition) has no native or FFI layer: its callers
 * import Python symbols.  Each entry point below therefore cites the Python interface whose work
 * it performs; the Python package `face_recognition_amd` re-creates those interfaces (same names,
 * arguments, return types) on top of this ABI through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a non-zero fra_status otherwise; fra_last_error()
 *     returns a thread-local, NUL-terminated description of the last failure on this thread;
 *   - no C++ exceptions cross this boundary, no torch types appear in it;
 *   - "dev" pointers are device memory owned by the caller and must stay valid until the work
 *     queued on `stream` has completed; all launches go to the caller's stream and return
 *     without synchronising unless stated otherwise;
 *   - a handle may be used from one thread at a time.
 */
#ifndef FACE_RECOGNITION_AMD_H
#define FACE_RECOGNITION_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fra_stream;           /* a hipStream_t (NULL = the default stream) */
typedef struct mfn_handle mfn_handle;
typedef struct fb_handle fb_handle;

enum fra_status {
    FRA_OK = 0, FRA_ERR_ARG = 1, FRA_ERR_STATE = 2, FRA_ERR_HIP = 3, FRA_ERR_WEIGHTS = 4, FRA_ERR_WORKSPACE = 5
};
enum fra_dtype { FRA_F32 = 0, FRA_BF16 = 1, FRA_F16 = 2 };

/* One entry of a state_dict: key name as the reference module registers it
 * (e.g. "conv_3.model.0.conv_dw.bn.running_var"), host pointer to float32 data, shape. */
typedef struct {
    const char* name;
    const float* data;      /* host memory, contiguous, float32 */
    int32_t ndim;
    int64_t shape[4];
} mfn_tensor_desc;

const char* fra_last_error(void);
int fra_version(void);

/* Kernels whose waves synchronise through flags in on-chip memory poll with a bound, so that a fault in that protocol
 * cannot hang the GPU; a poll that runs out marks the device, and the launch's results are wrong.  mfn_forward*,
 * mfn_op_block / mfn_op_rblock's callers and mfn_destroy report a mark left by EARLIER launches on their device as
 * FRA_ERR_STATE (fra_last_error names the kernel); fra_device_fault reads and clears the mark itself: returns
 * FRA_ERR_STATE and the tag (kernel letters << 16 | wave) if one was set.  Synchronise the stream first to cover the
 * launches still in flight.  No counterpart in the reference. */
int fra_device_fault(int device, unsigned* tag_out);
/* How batches of up to about 36 images (7 B <= compute units) run conv_4 / conv_5: the path behind the per-frame loop of
 * cam_demo.py:87-105.  Mode 1 (default): two 64-channel slices of a residual block at a time, the block's project sum added
 * in two halves -- the shortest launches; such a batch's embeddings equal those of a larger batch to rounding (cosine
 * 0.99999 on weights whose embeddings spread over the sphere), and are the same bits for every batch size inside the regime.  Mode 0: one slice at a time, the arithmetic of
 * the large-batch kernels value for value: an embedding is then the same bits whatever batch it is computed in.  Process
 * wide; takes effect at the next forward (graphs captured before keep the mode they were captured in).  No counterpart in
 * the reference. */
int fra_set_small_batch_mode(int mode);
int fra_get_small_batch_mode(void);
/* Test hook: how many times the library has called a memory allocator (hipMalloc, hipFree, hipHostMalloc) since it was
 * loaded.  Every such call in the library goes through one counted wrapper, so a count that does not move across a
 * call proves that the call allocated and freed nothing. */
long fra_dbg_alloc_calls(void);
/* Test hook: polls after which such a wait gives up (negative = the built-in bound of about 0.1 s). */
int fra_dbg_set_spin_limit(int polls);

/* ---- MobileFaceNet embedding forward ----------------------------------------------------------
 * Replaces MobileFaceNet.__init__/forward: Mobilefacenet/face_model.py:77-129. */

/* Create a model context on HIP device `device`.  act_dtype: FRA_F16 or FRA_BF16 storage of
 * activations and weights (arithmetic accumulates in fp32).  face_model.py:78 (embedding_size). */
int mfn_create(mfn_handle** out, int device, int embedding_size, int act_dtype);

/* Load a state_dict (the 333 keys of face_model.py:80-92; "num_batches_tracked" entries may be
 * omitted; a leading "module." is ignored).  Folds every BatchNorm into the preceding conv /
 * linear layer (eps 1e-5), repacks to the kernels' layouts and uploads.  Synchronous; a RE-load first waits for
 * the device to drain, since forwards in flight on any stream still read the old weights.
 * Replaces Module.load_state_dict as used at cam_demo.py:52, Evaluation.py:110, facebank.py:88. */
int mfn_load_weights(mfn_handle* h, const mfn_tensor_desc* tensors, int n);

/* Bytes of device workspace mfn_forward needs for batches up to max_batch. */
int mfn_workspace_bytes(const mfn_handle* h, int max_batch, size_t* bytes);

/* Process the batch in slices of `images` (0 = whole batch at once).  Smaller slices keep the
 * layer-to-layer activations inside the 256 MB Infinity Cache. */
int mfn_set_batch_chunk(mfn_handle* h, int images);

/* fused != 0 (the default): every inverted-residual block (expand, depthwise, project, shortcut) runs
 * as one kernel; 0: one kernel per conv.  Changes mfn_op_count / mfn_op_info / mfn_debug_prefix. */
int mfn_set_fused(mfn_handle* h, int fused);

/* chain != 0 (the default): a run of identical residual blocks (conv_4.model.0-5 and conv_5.model.0-1,
 * face_model.py:85-87) goes out as ONE launch that keeps the activation in registers between blocks, for batches
 * where whole-image workgroups fill the chip, and conv_4's blocks go out two per launch for small batches (bands that
 * compute their neighbours' nearest rows themselves); 0: one launch per block.  Same results bit for bit either way. */
int mfn_set_chain(mfn_handle* h, int chain);
/* first_op[i] (host, n_ops >= mfn_op_count entries) = the op whose launch covers op i in a batch of B images:
 * i itself, or the first block of the chain i runs in.  mfn_forward_timed books a chain's time on that op. */
int mfn_launch_plan(const mfn_handle* h, int B, int* first_op, int n_ops);

/* x: dev f32 [B,3,112,112] NCHW (BGR, [-1,1]) -> out: dev f32 [B,embedding_size], rows unit L2 norm.
 * Replaces MobileFaceNet.forward: face_model.py:103-129. */
int mfn_forward(mfn_handle* h, const float* x_nchw, int B, float* out, void* workspace, size_t workspace_bytes,
                fra_stream stream);

/* Extended forward.  input_format 0: x is f32 [B,3,112,112] NCHW in [-1,1] (as mfn_forward);
 * input_format 1: x is uint8 [B,112,112,3] HWC BGR pixels as cv2.imread / Face_alignment produce them, and
 * ToTensor + Normalize([.5]*3,[.5]*3) (facebank.py:22-24, cam_demo.py:83-85) is applied inside the first kernel.
 * tta != 0: out[b] = l2_norm(net(x[b]) + net(hflip(x[b]))), the flip test-time augmentation of
 * facebank.py:59-62 / cam_demo.py:88-92 / Evaluation.py:50-52, with the mirrored read done inside the
 * first kernel (no second input tensor); needs mfn_workspace_bytes_ex(..., tta=1).  Fused schedule only. */
int mfn_forward_ex(mfn_handle* h, const void* x, int input_format, int B, int tta, float* out, void* workspace,
                   size_t workspace_bytes, fra_stream stream);
int mfn_workspace_bytes_ex(const mfn_handle* h, int max_batch, int tta, size_t* bytes);

/* Same forward, bracketing every kernel with events on `stream`; synchronises; op_ms[i] receives
 * the milliseconds of op i (see mfn_op_count / mfn_op_info).  For benchmarks only. */
int mfn_forward_timed(mfn_handle* h, const float* x_nchw, int B, float* out, void* workspace,
                      size_t workspace_bytes, fra_stream stream, float* op_ms, int n_ops);
int mfn_op_count(const mfn_handle* h);
/* Which device kernel runs op `op` in a batch of B images (for benchmark labels; no counterpart in the reference):
 * 0 = the kernel of its kind (block_kernel for a fused block), 1 = rblock_kernel (residual blocks whose activation stays in
 * registers), 2 = pblock_kernel (the block as a pipeline of pixel tiles inside a workgroup).  Negative: an error code. */
int mfn_op_variant(const mfn_handle* h, int op, int B);
/* kind: 0 stem, 1 depthwise 3x3, 2 pointwise, 3 gdconv 7x7, 4 linear+l2norm, 5 fused block, 6 fused stem + first depthwise,
 * 7 fused tail (conv_6_sep + conv_6_dw).  flops / bytes are the
 * algorithmic work and HBM traffic (activations read + written) of the op for ONE image; weight_bytes
 * are its parameters, read once per launch whatever the batch. */
int mfn_op_info(const mfn_handle* h, int op, char* name, int name_cap, int* kind, double* flops_per_image,
                double* bytes_per_image, double* weight_bytes);

/* Debug tap: run the first n_ops ops on x and write that op's output activation, converted to
 * f32, NHWC, to out_nhwc_f32 (dev).  *numel_per_image receives its per-image element count. */
int mfn_debug_prefix(mfn_handle* h, const float* x_nchw, int B, int n_ops, float* out_nhwc_f32,
                     int64_t* numel_per_image, void* workspace, size_t workspace_bytes, fra_stream stream);

int mfn_destroy(mfn_handle* h);

/* out = (a + b) / ||a + b||_2 row-wise, dev f32 [B,E]; b == NULL gives l2_norm(a).
 * Replaces l2_norm (face_model.py:19-22, axis=1) and the flip-TTA combine
 * l2_norm(emb + emb_mirror): facebank.py:62, cam_demo.py:92, Evaluation.py:51-52. */
int mfn_tta_combine(const float* a, const float* b, int B, int E, float* out, fra_stream stream);

/* ---- single kernels, exposed for unit tests (parameters already folded / in kernel layout) ---- */
int mfn_op_stem(int dtype, const float* x, const float* w27x64, const float* bias, const float* slope, void* out,
                int B, fra_stream stream);
int mfn_op_dw3x3(int dtype, const void* in, const void* w9xC, const float* bias, const float* slope_or_null,
                 void* out, int B, int H, int W, int C, int stride, fra_stream stream);
int mfn_op_pw(int dtype, const void* in, const void* wNxK, const float* bias, const float* slope_or_null,
              const void* resid_or_null, void* out, int64_t M, int K, int N, fra_stream stream);
/* Fused inverted-residual block (face_model.py:55-65).  Arrays indexed by expanded channel on the
 * depthwise / project side (b1p, s1p, w2p [9][cexp], b2p, s2p) are in the kernel's permuted channel
 * order: natural channel 16g+8a+4h+i is stored at POSITION 16g+8h+4a+i.  w3p holds the cout x cexp
 * project weights slice-major in the order the project stage reads them: with pos the position of the
 * expanded channel, c = pos/64, P = (pos/32)%2, hi = (pos/16)%2, g = (pos/4)%4, i = pos%4, element
 * (co, pos) sits at ((c*2*cout + 32*(co/16) + 16*(g%2) + co%16)*4 + 2*P + g/2)*8 + 4*hi + i.
 * s1p / s2p hold (PReLU slope - 1): the kernel evaluates v + (slope-1)*min(v,0). */
int mfn_op_block(int dtype, int cin, int cexp, int cout, int h, int stride, int residual, const void* x,
                 const void* w1, const float* b1p, const float* s1p, const void* w2p, const float* b2p,
                 const float* s2p, const void* w3p, const float* b3, void* out, int B, fra_stream stream);
/* Residual blocks (stride 1, c -> cexp -> c, + x; face_model.py:55-75) whose weights travel as per-slice packages
 * in the kernel's LDS layout.  mfn_pack_rblock builds the packages of ONE block on the host from BN-folded
 * weights in natural order: w1 [cexp][c], w2 [cexp][9] (tap = 3 ky + kx), w3 [c][cexp], biases, PReLU slopes
 * s1 / s2 [cexp]; out must hold mfn_rblock_package_bytes(c, cexp) bytes (0 = unsupported widths).
 * mfn_op_rblock runs nblk blocks whose packages lie back to back at `packages` (dev) on x -> out, both
 * T NHWC [B][h][h][c] (h = 14 or 7).  nblk > 1 keeps the activation in registers between blocks, and out may then
 * be x (a workgroup owns whole images; two blocks of a small batch of 14 x 14 maps run as bands when out is not x);
 * with nblk == 1 small batches run bands that read their neighbours' rows, so out must not be x. */
int64_t mfn_rblock_package_bytes(int c, int cexp);
int mfn_pack_rblock(int dtype, int c, int cexp, const float* w1, const float* b1, const float* s1, const float* w2,
                    const float* b2, const float* s2, const float* w3, const float* b3, void* packages_host);
int mfn_op_rblock(int dtype, int c, int cexp, int h, const void* x, const void* packages, int nblk, void* out, int B,
                  fra_stream stream);
/* The same chain on the wide-wave kernel (four waves per workgroup, one per SIMD, four image rows each; h = 14 only):
 * what mfn_op_rblock itself launches for whole-image workgroups unless FRA_WBLOCK=0 is set.  Same packages, same
 * results bit for bit; out may be x for any nblk. */
int mfn_op_wblock(int dtype, int c, int cexp, int h, const void* x, const void* packages, int nblk, void* out, int B,
                  fra_stream stream);
int mfn_op_gdconv7(int dtype, const void* in, const void* w49xC, const float* bias, void* out, int B, int C,
                   fra_stream stream);
/* conv_6_sep + conv_6_dw in one kernel (face_model.py:88-89): x [B][49][128] NHWC, w6 [512][128] (BN folded),
 * b6 / s6 f32 [512] with s6 = PReLU slope - 1, w7 [49][512] tap-major, b7 f32 [512]; out [B][512]. */
int mfn_op_tail(int dtype, const void* x, const void* w6, const float* b6, const float* s6_minus_1, const void* w7,
                const float* b7, void* out, int B, fra_stream stream);
int mfn_op_linear_l2norm(int dtype, const void* in, const void* wExK, const float* bias, float* out, int B, int K,
                         int E, fra_stream stream);

/* ---- facebank: 1:N matching and top-k search -----------------------------------------------------
 * Replaces the inline match of the demos (cam_demo.py:96-105, Video_demo.py:99-108,
 * MTCNN_MobileFaceNet.py:76-85) on the tensor load_facebank returns (facebank.py:79-83). */

/* Build a gallery on `device` from N rows of D floats (host or device memory; `gallery_is_device`
 * says which).  store_dtype FRA_F32 keeps rows in fp32 for the exact small-facebank match;
 * FRA_F16 / FRA_BF16 pack them for the MFMA search and precompute ||g||^2 from the packed values.
 * index_base is added to every index reported (row offset of this shard in the whole gallery). */
int fb_create(fb_handle** out, int device, const float* gallery, int gallery_is_device, int64_t N, int D,
              int store_dtype, int64_t index_base);
/* Write rows [row0, row0+n) of a gallery created with gallery == NULL (or overwrite rows); lets a
 * large shard be produced piecewise on the device.  Synchronous; runs on a stream of the handle's own, so device
 * rows must be complete when it is called (synchronise their producer stream first) and no other stream is stalled --
 * which also means that searches still in flight on other streams read rows while they change: let them finish first
 * (the Python Gallery.fill synchronises the caller's stream before host rows are written). */
int fb_fill(fb_handle* h, int64_t row0, const float* rows, int rows_is_device, int64_t n);
/* fb_fill for device rows, queued on `stream` and not waited for: searches queued on the same stream afterwards see
 * the rows; rows_dev must stay valid until the stream reaches that point. */
int fb_fill_async(fb_handle* h, int64_t row0, const float* rows_dev, int64_t n, fra_stream stream);
int64_t fb_rows(const fb_handle* h);
/* Packed form of rows [row0, row0+n) of a 16-bit gallery: the 16-bit rows [n,D] and the fp32 bias terms
 * -|g|^2/2 [n] exactly as the search kernel reads them.  Export after fb_fill and import into a gallery
 * of the same store dtype reproduce every search result bit for bit and skip the fp32 -> 16-bit
 * conversion; this is what a gallery file on disk holds (the reference keeps its facebank as one fp32
 * tensor in facebank.pth, facebank.py:72-75,79-83).  Host or device pointers; synchronous. */
int fb_export_packed(fb_handle* h, int64_t row0, int64_t n, void* rows16_out, int rows_is_device, float* negh_out,
                     int negh_is_device);
int fb_import_packed(fb_handle* h, int64_t row0, int64_t n, const void* rows16, int rows_is_device, const float* negh,
                     int negh_is_device);

/* Pre-size the handle's internal scratch for searches of up to max_queries queries, so that no
 * search of that size allocates (e.g. before capturing searches into a hipGraph).  Scratch only
 * grows, and blocks already used are kept until fb_destroy, so captured graphs stay valid. */
int fb_reserve(fb_handle* h, int max_queries);

/* dist_out[q,j], idx_out[q,j] (dev, [Q,k]) = the k smallest squared L2 distances
 * sum_d (q[d]-g[n,d])^2 and their (global) indices, ascending; equal distances -> lowest index.
 * Slots beyond the gallery size hold distance +inf and index -1.  1 <= k <= 8.
 * Generalises  minimum, min_idx = torch.min(dist, dim=1)  of cam_demo.py:97-99. */
int fb_search_l2_topk(fb_handle* h, const float* queries, int Q, int k, float* dist_out, int32_t* idx_out,
                      fra_stream stream);

/* fb_search_l2_topk for callers that cannot allocate (code under hipGraph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same results bit
 * for bit, but the scratch the search kernels need (16-bit queries, |q|^2, per-split partial top-k lists;
 * for fp32 galleries the Q x N distance matrix) lives in an 8 MiB array that is part of the library's
 * device image, one copy per GPU, kept until the process ends; the call allocates and frees nothing.
 * A search that needs more than 8 MiB of scratch is refused with FRA_ERR_ARG (the message states the
 * bytes needed): 16-bit galleries take up to 256 queries at any k and any gallery size (512 at k <= 5);
 * fp32 galleries take Q * N <= 2,097,152.  Nothing in the array has to start at zero.  All calls on one
 * GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race. */
int fb_search_l2_topk_noalloc(fb_handle* h, const float* queries, int Q, int k, float* dist_out, int32_t* idx_out,
                              fra_stream stream);

/* Merge R lists of k candidates per query (dist_in / idx_in dev [R,Q,k], e.g. the result of an
 * all-gather of per-shard fb_search_l2_topk outputs) into the k best, same ordering rules. */
int fb_merge_topk(const float* dist_in, const int32_t* idx_in, int R, int Q, int k, float* dist_out,
                  int32_t* idx_out, fra_stream stream);

/* The demos' match step in one call (cam_demo.py:96-105): for each query the squared L2 distance to the nearest row of
 * an fp32 gallery (minimum dev f32 [Q]), that row's index as int64 -- -1 where minimum > (threshold-156)/(-80) -- and
 * clamp(-80*minimum+156, 0, 100) (score_100 dev f32 [Q]); the values of fb_search_l2_topk(k = 1) followed by
 * fb_threshold_score, in two kernel launches instead of four. */
int fb_match_nearest(fb_handle* h, const float* queries, int Q, double threshold, float* minimum, int64_t* idx_out,
                     float* score_100, fra_stream stream);

/* Demo post-processing on top-1 results (cam_demo.py:100,105): idx_io[q] = -1 where
 * minimum[q] > (threshold-156)/(-80); score_100[q] = clamp(-80*minimum[q]+156, 0, 100). */
int fb_threshold_score(const float* minimum, int64_t* idx_io, float* score_100, int Q, double threshold,
                       fra_stream stream);

int fb_destroy(fb_handle* h);

/* ---- verification protocol on the device -------------------------------------------------------
 * One fold of evaluation_10_fold (Mobilefacenet/Evaluation.py:83-94):
 *     mu = mean(concat(fl[val], fr[val]), axis 0);  fl -= mu; fr -= mu;
 *     fl /= |fl| (row-wise);  fr /= |fr|;
 *     scores = sum((fl - fr)^2, axis 1)   (method 0, "l2_distance")
 *            = sum(fl * fr, axis 1)       (method 1, "cos_distance")
 * fl, fr: dev f32 [P,D], UPDATED IN PLACE -- the reference re-assigns the centred, normalised
 * arrays inside its fold loop (:86-89), so the centring accumulates from fold to fold; call this once
 * per fold on the same buffers.  val_mask: dev uint8 [P], 1 = pair belongs to the folds the mean is
 * taken over (n_val of them).  mu_scratch: dev f32 [D].  scores_out: dev f32 [P] (all pairs).
 * Every float32 operation is issued in numpy's order, so the scores equal the reference's bit for bit.
 * D in {128, 256, 512}. */
int ev_fold_scores(float* fl, float* fr, const uint8_t* val_mask, int n_val, int P, int D, int method,
                   float* mu_scratch, float* scores_out, fra_stream stream);

/* ---- face alignment warp ------------------------------------------------------------------------
 * Mobilefacenet/utils/align_trans.py:50-52 (and :65-66 with a 96-wide crop):
 *     cv2.warpAffine(img, M[:2], (W, H))[0:out_h, 0:out_w]        for every detected face
 * frame: dev uint8 [H,W,3] (the BGR image as cv2 holds it).  matrices: dev f64 [n_faces][6], the
 * forward 2x3 matrices M[:2] row-major, as transformation_from_points (:16-29) returns them.
 * crops: dev uint8 [n_faces,out_h,out_w,3] -- with out_h = out_w = 112 exactly the batch
 * mfn_forward_ex(input_format = 1) reads.  Bilinear, constant 0 border, OpenCV's 8-bit fixed-point
 * arithmetic (1/32-pixel coordinates, 15-bit weights).  out_w must be a multiple of 4. */
int fa_align_warp(const uint8_t* frame, int H, int W, const double* matrices, int n_faces, uint8_t* crops,
                  int out_h, int out_w, fra_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* FACE_RECOGNITION_AMD_H */
