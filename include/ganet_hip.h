/*
 * ganet_hip.h - C-ABI of the MI355X (gfx950) implementation of GANet's hot path:
 *   (1) the Lane-aware Feature Aggregator's deformable gather-and-aggregate
 *       (DeformConv / DeformConv1D / ModulatedDeformConv forward), and
 *   (2) the global keypoint-association decode (heat-map NMS, threshold,
 *       ordered compaction, start-point regression, lane grouping).
 *
 * Plain C: pointers, sizes and PODs only - no torch / HIP types.  Every device
 * pointer is caller-owned; the library never allocates or frees device memory
 * in these calls (scratch comes in through the `workspace` arguments), never
 * synchronises with the host, and enqueues only on the stream it is given.
 * All entry points are re-entrant; the only state the library keeps between calls
 * is the per-GPU address of the static scratch array of ganet_decode_batch_noalloc.
 * A few measurement knobs are read from the environment ("Build, limits");
 * those that size a workspace are read once per process: set them before the first call.
 * Call them from one host thread per GPU with that GPU current.
 * Forward AND backward of the operators are here (ganet_deform_conv_backward), plus
 * host-only routines that turn decoded records into the reference's result formats.
 *
 * Citations are file:line in the reference tree (Wolfwjs/GANet).
 */
#ifndef GANET_HIP_H_
#define GANET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GANET_HIP_ABI_VERSION 1

/* hipStream_t, passed opaquely (NULL = the default stream). */
typedef void *ganet_stream_t;

typedef enum {
  GANET_OK = 0,
  GANET_ERR_NULL = -1,        /* a required pointer is NULL */
  GANET_ERR_BAD_SHAPE = -2,   /* shapes the reference's shape_check rejects */
  GANET_ERR_UNSUPPORTED = -3, /* valid for the reference, not built here yet */
  GANET_ERR_WORKSPACE = -4,   /* workspace_bytes smaller than the query says */
  GANET_ERR_HIP = -5,         /* a HIP call / kernel launch failed */
  GANET_ERR_TOO_SMALL = -6,   /* convolution output would be empty */
  GANET_ERR_NOT_FINITE = -7   /* a coordinate Python's int() would refuse (NaN, infinity)
                                 or that no int32 holds (host formatting calls) */
} ganet_status;

typedef enum { GANET_F32 = 0, GANET_BF16 = 1, GANET_F16 = 2 } ganet_dtype;
typedef enum { GANET_NCHW = 0, GANET_NHWC = 1 } ganet_layout;

const char *ganet_status_string(int status);
int ganet_abi_version(void);

/* ------------------------------------------------------------------------
 * (1) Deformable convolution forward.
 *
 * Replaces the pybind entry point
 *   int deform_conv_forward(Tensor input, Tensor weight, Tensor offset,
 *       Tensor output, Tensor columns, Tensor ones, int kW, int kH, int dW,
 *       int dH, int padW, int padH, int dilationW, int dilationH, int group,
 *       int deformable_group, int im2col_step)
 *   mmdet/ops/dcn/src/deform_conv_ext.cpp:51-67 -> deform_conv_forward_cuda,
 *   mmdet/ops/dcn/src/cuda/deform_conv_cuda.cpp:152-260,
 * and, with `mask` (and optionally `bias`) non-NULL,
 *   modulated_deform_conv_forward, deform_conv_ext.cpp:98-113 ->
 *   deform_conv_cuda.cpp:490-569.
 * It is called from DeformConvFunction.forward (mmdet/ops/dcn/deform_conv.py:52-57).
 *
 * There is no im2col buffer and no im2col_step chunking: sampling and the
 * contraction are one fused kernel, or (2*Cout <= Cin; 16-bit arithmetic, or float32
 * tensors contracted as three bf16 passes) the commuted pair "per-tap GEMM, then
 * sampling" - see `algo` below.
 * ---------------------------------------------------------------------- */
typedef struct {
  int32_t B, Cin, H, W, Cout;
  int32_t kH, kW;           /* weight.size(2), weight.size(3) */
  int32_t sH, sW;           /* stride   ("dH, dW" in the reference signature) */
  int32_t pH, pW;           /* padding */
  int32_t dH, dW;           /* dilation */
  int32_t groups;           /* conv groups */
  int32_t deformable_groups;
  int32_t dtype;            /* ganet_dtype of x, weight, (bias,) out.  offset and
                               mask are always float32. */
  int32_t x_layout;         /* ganet_layout of x */
  int32_t out_layout;       /* ganet_layout of out */
  int32_t mfma_bf16;        /* dtype == GANET_F32 only, how float32 tensors are
                               contracted:
                               0 = float32-class accuracy (default): three bf16
                                   MFMA passes over hi/lo splits of both operands
                                   with float32 accumulation (algorithm 2 splits x,
                                   algorithm 1 its blended tile; <= 2^-15 relative
                                   per product, 4e-6..8e-6 of the output scale;
                                   non-finite inputs give NaN);
                               1 = round operands to bf16, one bf16 MFMA pass
                                   (~1e-2 of the output scale);
                               2 = exact float32 MFMA only */
  int32_t fuse_relu;        /* 1 = out = max(out, 0)
                               (mmdet/models/dense_heads/lanepoints_conv.py:172) */
  int32_t offset_is_points; /* 1 = `offset` holds the LFA's pts_out_init and the
                               kernel subtracts the kernel-grid base itself
                               (lanepoints_conv.py:63-74,170); needs kH == 1,
                               dW == 1, pW == (kW-1)/2 */
  int32_t weight_is_packed; /* 1 = `weight` was produced by
                               ganet_deform_conv_pack_weight for these params
                               (inference: pack once, not per call) */
  int32_t algo;             /* 0 = choose automatically; 1 = sample-then-contract
                               (one fused gather+blend+MFMA kernel);
                               2 = contract-then-sample (per-tap GEMM into a
                               cache-resident scratch, then sampling; needs one
                               deformable group, stride 1, 2*Cout <= Cin);
                               3 = algorithm 2 as ONE persistent launch (16-bit
                               arithmetic, Cout <= 64 per tap, "same" output
                               size; opt-in, never chosen automatically:
                               slower than 2 on MI355X so far);
                               4 = algorithm 2 as ONE launch with the per-tap
                               products kept in the forming XCD's L2 (formed tap by
                               tap into a three-plane ring that is overwritten in
                               place; 16-bit arithmetic, Cin 64/128/256, Cout <= 64,
                               "same" output size, <= 16 taps, H*W <= 4096; needs
                               the 8 x 32-CU layout of MI355X);
                               5 = small batches (the reference's speed test runs
                               batch 1, tools/ganet/speed_test.py:57-65): ONE launch in
                               which a workgroup owns 16 output pixels x 64 outputs and
                               its wavefronts split the taps; sample-then-contract like
                               1, partial sums added in a fixed order.  One deformable
                               group; float32 tensors with mfma_bf16 0 or 1.  Chosen
                               automatically while a call is one round of workgroups
                               (B*Hout*Wout/16 * ceil(Cout/64) <= 256; process-wide
                               knob GANET_TAPS_MAX_TILES, 0 = never).
                               | GANET_DCN_ALGO_KEEP_Y: training - see below */
  int32_t debug;            /* must be 0 (ablation bits of profiling builds) */
} ganet_dcn_params;

/* Added to `algo` (0 or 2) of a FORWARD call: the per-tap products Y = x * W_t^T of the
 * whole batch, float32 [B][H*W][kH*kW*Cout], are left in the workspace at
 * ganet_deform_conv_y_offset(p) (one chunk instead of cache-sized ones: the forward gets a
 * little slower and the workspace holds B*H*W*kH*kW*Cout*4 bytes), for
 * ganet_deform_conv_backward_y, which then does not form them again.  float32 tensors in the
 * default arithmetic (mfma_bf16 = 0) where algorithm 2 applies and Cout % 64 == 0;
 * GANET_ERR_UNSUPPORTED otherwise (call again without the bit).  The reference keeps nothing
 * between its forward and backward either way (deform_conv.py:52-57, :60-96 recompute the
 * columns); this is an addition for training throughput. */
#define GANET_DCN_ALGO_KEEP_Y 0x100

/* Added to `algo` of a BACKWARD call: grad_x is written instead of added into, so the
 * caller need not zero-fill it (the reference's Python does, deform_conv.py:73-75, and its
 * col2im adds).  Where the gradient comes straight out of a GEMM (commuted form, float32
 * channel-last tensors) this saves the fill and the read of the old values; elsewhere the
 * library clears the tensor itself.  grad_weight / grad_bias are still added into. */
#define GANET_DCN_BWD_OVERWRITE_X 0x200

/* Output spatial size rule of the reference
 * (mmdet/ops/dcn/deform_conv.py:98-111, deform_conv_cuda.cpp:108-111). */
int ganet_dcn_out_size(int in, int pad, int dilation, int k, int stride);

/* Bytes of device scratch `ganet_deform_conv_forward` needs for `p`
 * (channel-last copy of x when x_layout is NCHW, repacked weights, the
 * per-tap products of the contract-then-sample algorithm).
 * Returns 0 for invalid parameters. */
size_t ganet_deform_conv_workspace_bytes(const ganet_dcn_params *p);

/* Which algorithm ganet_deform_conv_forward runs for `p`: 1, 2, 3, 4 or 5 as in
 * ganet_dcn_params.algo (0 for invalid parameters).  No counterpart in the
 * reference; for logs and benchmarks. */
int ganet_deform_conv_algorithm(const ganet_dcn_params *p);

/* Algorithms 3 and 4 synchronise their workgroups through counters in the workspace; every wait
 * is bounded, and a wait that runs out (a launch the GPU could not keep resident) leaves
 * a non-zero 32-bit word in the workspace instead of hanging: the output is then invalid.
 * Returns that word's byte offset inside `workspace` for `p`, or (size_t)-1 when the
 * algorithm chosen for `p` has no such word.  The word is valid once the stream has
 * finished the call.  No counterpart in the reference (its launches cannot fail this way). */
size_t ganet_deform_conv_status_offset(const ganet_dcn_params *p);

/* Byte offset of the kept per-tap products inside `workspace` for a forward call with
 * GANET_DCN_ALGO_KEEP_Y, (size_t)-1 when `p` does not keep them. */
size_t ganet_deform_conv_y_offset(const ganet_dcn_params *p);

/* Weight pre-packing (no counterpart in the reference, which hands the raw
 * [Cout,Cin,kH,kW] tensor to cuBLAS on every call, deform_conv_cuda.cpp:228-236):
 * the kernels read weights as [tap][Cout][Cin] in their arithmetic dtype (float32 tensors
 * in the default arithmetic: a hi and a lo bf16 plane).  16-bit planes are followed by a
 * second, fragment-major copy of the same values (blocks of 16 outputs x 32 channels in
 * MFMA operand order) that algorithm 5 reads; both are written whatever the batch size of
 * `p`, so a buffer packed for a shape serves it at every batch size.
 * ganet_deform_conv_forward packs into its workspace on every call unless
 * p->weight_is_packed is set and `weight` points at a buffer filled here. */
size_t ganet_deform_conv_packed_weight_bytes(const ganet_dcn_params *p);
int ganet_deform_conv_pack_weight(const void *weight, const ganet_dcn_params *p,
                                  void *packed, size_t packed_bytes,
                                  ganet_stream_t stream);

/* x      [B,Cin,H,W]  (or NHWC bytes when x_layout == GANET_NHWC), p->dtype
 * offset [B, 2*dg*kH*kW, Hout, Wout] float32, channel 2t = dy, 2t+1 = dx of tap t
 *        (deform_conv_cuda_kernel.cu:222-225)
 * mask   NULL, or [B, dg*kH*kW, Hout, Wout] float32 (DCNv2)
 * weight [Cout, Cin/groups, kH, kW], p->dtype
 * bias   NULL, or [Cout], p->dtype (DCNv2 only)
 * out    [B,Cout,Hout,Wout] (or NHWC), p->dtype; fully overwritten.
 * Returns GANET_OK after enqueueing on `stream`; no host synchronisation. */
int ganet_deform_conv_forward(const void *x, const float *offset,
                              const float *mask, const void *weight,
                              const void *bias, void *out,
                              const ganet_dcn_params *p, void *workspace,
                              size_t workspace_bytes, ganet_stream_t stream);

/* Backward (training), float32, conv groups == 1.
 *
 * Replaces deform_conv_backward_input_cuda + deform_conv_backward_parameters_cuda
 * (mmdet/ops/dcn/src/cuda/deform_conv_cuda.cpp:262-488; called from
 * DeformConvFunction.backward, mmdet/ops/dcn/deform_conv.py:60-96) and, with a mask,
 * modulated_deform_conv_cuda_backward (deform_conv_cuda.cpp:571-700;
 * ModulatedDeformConvFunction.backward, deform_conv.py:152-172).
 *
 * x, offset, mask, weight as in the forward call; grad_out has the forward output's
 * shape and p->out_layout.  Like the reference, which adds into tensors its Python
 * side zero-fills, grad_x [like x], grad_weight [like weight] and grad_bias [Cout] are
 * ACCUMULATED INTO; grad_offset [like offset] and grad_mask [like mask] are overwritten.
 * grad_x and grad_offset are computed together (both NULL to skip them, as when only
 * the weight needs a gradient); grad_mask is required with them when mask != NULL;
 * grad_weight and grad_bias may each be NULL.  There is no `columns` buffer and no
 * im2col_step.  p->algo picks the form: 1 = direct (the columns are formed and consumed
 * on chip), 2 = commuted (Cout-wide rows scattered / gathered through the per-tap
 * products Y_t = X * W_t^T, grad_x and grad_weight as GEMMs; one deformable group,
 * stride 1, Cout and the padded Cin multiples of 64, else GANET_ERR_UNSUPPORTED),
 * 0 = commuted where it applies and 2 * Cout <= Cin.  Unsupported
 * (GANET_ERR_UNSUPPORTED): other dtypes, conv groups > 1, deformable groups whose
 * channel count is not a multiple of 32, fuse_relu / offset_is_points /
 * weight_is_packed. */
size_t ganet_deform_conv_backward_workspace_bytes(const ganet_dcn_params *p);
int ganet_deform_conv_backward(const void *x, const float *offset, const float *mask,
                               const void *weight, const void *grad_out, void *grad_x,
                               float *grad_offset, float *grad_mask, void *grad_weight,
                               void *grad_bias, const ganet_dcn_params *p,
                               void *workspace, size_t workspace_bytes,
                               ganet_stream_t stream);

/* The same backward with the forward's per-tap products handed over (`y_saved` = the
 * forward's workspace + ganet_deform_conv_y_offset, x / offset / weight unchanged since):
 * the commuted form skips its own contraction of x (about a sixth of the call at the LFA
 * shape).  The products are the ones ganet_deform_conv_backward forms itself (the same
 * three bf16 passes), so the gradients are the same up to the order of the atomic sums.  float32 tensors,
 * commuted form only (GANET_ERR_UNSUPPORTED otherwise). */
int ganet_deform_conv_backward_y(const void *x, const float *offset, const float *mask,
                                 const void *weight, const void *grad_out, void *grad_x,
                                 float *grad_offset, float *grad_mask, void *grad_weight,
                                 void *grad_bias, const ganet_dcn_params *p,
                                 void *workspace, size_t workspace_bytes,
                                 ganet_stream_t stream, const void *y_saved);

/* ------------------------------------------------------------------------
 * (2) Keypoint decode + lane grouping for a batch of images.
 *
 * Replaces, per image,
 *   GANetHeadFast.ktdet_decode_fast(heat, offset, error, thr, root_thr)
 *     mmdet/models/dense_heads/ganet_head.py:203-238
 *   group_points_fast(kpt_seeds, cpt_seeds, thr, by_center_thr)
 *     tools/ganet/post_process.py:141-194   (PostProcessor.__call__ :244-262)
 * The reference handles one image per call; this entry point takes the whole
 * batch in one launch sequence.
 * ---------------------------------------------------------------------- */
#define GANET_DECODE_MAX_CENTERS 2048 /* hard limit of max_centers / max_groups
                                         in by-centre mode (kept in LDS) */
#define GANET_DECODE_COUNT_STRIDE 8

typedef struct {
  int32_t B, H, W;
  float kpt_thr;        /* `thr`: heat_nms > thr, compared in float32 */
  float root_thr;       /* offset_y < root_thr, compared in float32 */
  double cluster_thr;   /* PostProcessor.cluster_thr: used when by_center_thr < 0
                           (update_coords_fast, post_process.py:142-153) */
  double by_center_thr; /* PostProcessor.cluster_by_center_thr; < 0 means None.  Cost
                           note for None: update_coords_fast clusters the KEYPOINTS
                           greedily in order (post_process.py:142-153), which is serial
                           per image here too: nk^2/64 distance evaluations on one
                           wavefront - milliseconds from ~10^4 keypoints per image
                           (a negative kpt_thr selects every pixel).  By-centre mode,
                           which every shipped config uses, has no such term. */
  int32_t max_kpts;     /* capacity of the per-image keypoint arrays */
  int32_t max_centers;  /* capacity of the per-image centre arrays (<= 2048) */
  int32_t max_groups;   /* capacity of the per-image group arrays */
  int32_t flags;        /* bit0: skip grouping (decode only).
                           The reference's behaviour is flags bits 1-2 = 0; the two
                           variants below have no counterpart in its code (they are the
                           forms its paper / BASELINE.json's north star describe):
                           bit1: 3x3 heat-map NMS, max_pool2d(heat,(3,3),1,(1,1)),
                                 instead of the reference's 1x3 (ganet_head.py:205-208);
                           bit2: a keypoint joins the NEAREST group centre within the
                                 threshold (ties: lower index) instead of the first
                                 one (post_process.py:61-66)
                           bit3: `heat` holds the head's LOGITS; every value read goes
                                 through clamp(sigmoid(v), 1e-4, 1 - 1e-4) first, the line
                                 in front of the decode in forward_test
                                 (mmdet/models/dense_heads/ganet_head.py:291), so that pass
                                 over the plane and its launch go away.  float32,
                                 1 / (1 + exp(-v)) like ATen's GPU kernel; a sigmoid from
                                 another library can differ in the last bit, which only
                                 matters for a value within one ulp of kpt_thr or of a
                                 neighbour (checked bit-exact against torch's own sigmoid
                                 on this GPU, tests/test_gpu_decode.py) */
  int32_t reserved[4];
} ganet_decode_params;

/* Per-image outputs, device pointers, caller-owned, image b at index b*stride:
 *   counts       [B][8] int32: [0] keypoints found, [1] centres found,
 *                [2] groups, [3] overflow bits (1: keypoints > max_kpts,
 *                2: centres > max_centers, 4: groups > max_groups; arrays then
 *                hold a truncated, NOT reference-exact result), [4..7] reserved
 *   centers_xy   [B][max_centers][2] int32 (x,y), x-major order
 *                (= root_center_arr, ganet_head.py:217-219)
 *   center_gid   [B][max_centers] int32 group of each centre (by-centre mode)
 *   kpt_xy       [B][max_kpts][2] int32 (x,y), row-major order (ganet_head.py:231)
 *   kpt_align    [B][max_kpts][2] float32 coord + error  (ganet_head.py:230,233)
 *   kpt_root     [B][max_kpts][2] float32 coord + offset (ganet_head.py:229,232)
 *   kpt_gid      [B][max_kpts] int32 group index, -1 = in no group
 *   group_center [B][max_groups][2] int32 truncated mean centre
 *                (choose_mean_point, post_process.py:38-41); by-centre mode only
 *   group_count  [B][max_groups] int32 keypoints per group
 */
typedef struct {
  int32_t *counts;
  int32_t *centers_xy;
  int32_t *center_gid;
  int32_t *kpt_xy;
  float *kpt_align;
  float *kpt_root;
  int32_t *kpt_gid;
  int32_t *group_center;
  int32_t *group_count;
} ganet_decode_out;

size_t ganet_decode_workspace_bytes(const ganet_decode_params *p);

/* heat [B,1,H,W], offset [B,2,H,W] (ch0 = dx, ch1 = dy to the start point),
 * error [B,2,H,W], all float32 NCHW, contiguous.
 * Returns GANET_OK after enqueueing on `stream`; capacity overflow is reported
 * per image in `counts` (device memory), not in the return value. */
int ganet_decode_batch(const float *heat, const float *offset,
                       const float *error, const ganet_decode_params *p,
                       const ganet_decode_out *out, void *workspace,
                       size_t workspace_bytes, ganet_stream_t stream);

/* The same call for callers that cannot allocate device memory or carry a workspace
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * No counterpart in the reference.  The keypoint bitmap that `workspace` holds in
 * ganet_decode_batch lives in a static device array of the library instead
 * (GANET_DECODE_NOALLOC_MAX_BYTES per GPU, held until the process ends); the call
 * allocates and frees nothing.  Needs B * ceil(H*W/64) * 8 bytes (rounded up to 256)
 * <= GANET_DECODE_NOALLOC_MAX_BYTES, else GANET_ERR_WORKSPACE.  Every call on one GPU
 * shares that GPU's array: calls that overlap on one GPU (two streams, two threads)
 * race; run them one after the other on one stream.  Results are identical to
 * ganet_decode_batch.  The FIRST call on a GPU looks the array's address up
 * (hipGetSymbolAddress, which may load the code object): make one call per GPU outside any
 * graph capture before capturing it. */
#define GANET_DECODE_NOALLOC_MAX_BYTES (4u << 20)
int ganet_decode_batch_noalloc(const float *heat, const float *offset,
                               const float *error, const ganet_decode_params *p,
                               const ganet_decode_out *out, ganet_stream_t stream);

/* Grouping only, on lists that already sit in `out` on the device (written by
 * an earlier ganet_decode_batch with flags bit0, or uploaded by the caller):
 * reads counts[0..1], centers_xy and kpt_root; writes center_gid, kpt_gid,
 * group_center, group_count and counts[2..3].  This is the device side of
 *   PostProcessor.__call__(output, downscale)  tools/ganet/post_process.py:244-262
 * when decode and grouping are requested separately, as the reference's
 * callers do (tools/ganet/tusimple/test_dataset.py:193-196).
 * Only B, max_*, cluster_thr and by_center_thr of `p` are used. */
int ganet_group_batch(const ganet_decode_params *p, const ganet_decode_out *out,
                      ganet_stream_t stream);

/* Compacts the capacity-strided arrays of `out` into one dense int32 stream so
 * that the device-to-host copy of the lane lists moves only what was found
 * (the reference copies heat_nms-selected rows with three blocking .cpu()
 * calls, ganet_head.py:217,232-233).  Layout of `packed`:
 *   header[B][8]: n_kpts, n_centers, n_groups, overflow bits (bit 3: this record
 *                 did not fit in packed_words), record offset in words counted
 *                 from the end of the header, record words, total record words
 *                 (in header[0] only), 0
 *   record b    : kpt_xy[2nk] kpt_align[2nk] kpt_root[2nk] kpt_gid[nk]
 *                 centers_xy[2nc] center_gid[nc] group_center[2ngc] group_count[ng]
 *                 (counts clamped to the capacities; ngc = ng in by-centre mode,
 *                 0 otherwise; float arrays keep their bit patterns)
 * Records are laid out in image order.  packed_words >= 8*B. */
int ganet_decode_pack(const ganet_decode_params *p, const ganet_decode_out *out,
                      int32_t *packed, size_t packed_words, ganet_stream_t stream);

/* ganet_decode_batch and ganet_decode_pack as ONE call: `out` is filled as by
 * ganet_decode_batch and `packed` (device memory or host memory the device can write, e.g. a
 * pinned block) receives the record stream of ganet_decode_pack - header and record layout
 * as above, every record's words identical.  Replaces ktdet_decode_fast's three blocking
 * device-to-host copies (ganet_head.py:217,232-233) with stores that leave the kernel which
 * found the keypoints.  Where one workgroup per image does the whole decode (every image of at
 * most 8192 pixels, and the no-centre mode at any size) this is a single launch and the
 * records are laid out in the order the images finished sizing them, NOT in image order:
 * readers go by header[b][4..5] (ganet_lanes_from_packed does).  Larger by-centre images run
 * ganet_decode_batch's kernels followed by the pack kernel (image order).
 * Workspace: ganet_decode_packed_workspace_bytes(p) bytes, 8-byte aligned, ZERO-FILLED ONCE
 * before its first use (hipMemset): its last 256 bytes hold two counters that every call
 * leaves at zero again.  Calls that share a workspace must be ordered on one stream. */
size_t ganet_decode_packed_workspace_bytes(const ganet_decode_params *p);
int ganet_decode_batch_packed(const float *heat, const float *offset, const float *error,
                              const ganet_decode_params *p, const ganet_decode_out *out,
                              int32_t *packed, size_t packed_words, void *workspace,
                              size_t workspace_bytes, ganet_stream_t stream);

/* ---- after grouping: records -> the reference's result formats (host code) -----------
 * The reference runs these per image in Python (one interpreter step per point); here one
 * call formats a whole batch from the record stream of ganet_decode_pack on `threads` host
 * threads.  Nothing here touches the GPU; `packed` is host memory.
 *
 * ganet_lanes_from_packed =
 *   PostProcessor.lane_post_process   tools/ganet/post_process.py:216-242
 *     (groups of more than one point become lanes, points = align * downscale as float32,
 *      centers = root * downscale as float64, cluster centres = group_center * downscale)
 *   adjust_result   tools/ganet/tusimple/test_dataset.py:49-83 == culane/test_dataset.py:79-113
 *     (`int(v * ratio + offset)`, ratio = crop extent / network input extent, truncation
 *      towards zero; lanes with more than points_thr points are kept).
 * float32_scalars: how NumPy evaluates `np.float32 * python_float` for the points - 1 keeps
 * float32 (NumPy >= 2), 0 promotes to float64 (NumPy 1.x, the reference's own environment).
 * Layout of `lanes` (int32):
 *   header[B][4]: lanes kept, cluster centres, record offset in words counted from the end
 *                 of the header, record words
 *   record b    : per lane { id_class, n, points[2n] (x, y), centers[2n] } in group order,
 *                 then cluster_centers[2 * centres]
 * *needed_words is always set; GANET_ERR_WORKSPACE when lanes_words is smaller (lanes may be
 * NULL to query).  GANET_ERR_BAD_SHAPE for a record with capacity overflow bits set,
 * negative counts or an extent outside packed_words, GANET_ERR_NOT_FINITE where Python's
 * int() raises (or yields an integer no int32 holds).  Host memory running out inside any of
 * the three calls reads as GANET_ERR_WORKSPACE; no C++ exception leaves them. */
typedef struct ganet_lane_format {
  int32_t B;
  int32_t by_center;        /* records carry group centres (cluster_by_center_thr given) */
  int32_t downscale;        /* hm_down_scale (8 or 4) */
  int32_t points_thr;
  int32_t float32_scalars;
  int32_t threads;          /* host threads over images (<= 1: the calling thread) */
  int32_t img_h, img_w;     /* img_shape[:2], the network input */
  double crop_x0, crop_y0, crop_x1, crop_y1;   /* crop_bbox in original-image pixels */
  int64_t packed_words;     /* int32 words in `packed` (>= 8*B): a record that does not lie
                               inside is refused with GANET_ERR_BAD_SHAPE, not read */
} ganet_lane_format;

int ganet_lanes_from_packed(const int32_t *packed, const ganet_lane_format *f, int32_t *lanes,
                            size_t lanes_words, size_t *needed_words);

/* tusimple_convert_formal / get_line_intersection (tools/ganet/common.py:118-147) on the
 * stream of ganet_lanes_from_packed: x of every kept lane at each of the n_h rows of
 * h_samples (first segment whose y range holds the row, `int(k*(y-y0)+x0+0.49999)`),
 * reg_x (-2) where the lane has none.  lane_start[B+1]: first row of xs of every image;
 * xs[lane_start[B]][n_h].  GANET_ERR_WORKSPACE when xs_words is too small (lane_start is
 * filled all the same). */
int ganet_tusimple_from_lanes(const int32_t *lanes, int B, const int32_t *h_samples, int n_h,
                              int reg_x, int threads, int32_t *xs, size_t xs_words,
                              int32_t *lane_start);

/* out_result (tools/ganet/culane/test_dataset.py:138-148): the text of every image's
 * `.lines.txt` - one lane per line, "x y" pairs printed with two decimals.
 * text_start[2B+1]: [0..B] start of every image's slot in `text` (slot sizes are upper
 * bounds; text_start[B] = bytes needed), [B+1..2B] bytes used in every slot.
 * GANET_ERR_WORKSPACE when text_bytes < text_start[B] (text may be NULL to query). */
int ganet_culane_text(const int32_t *lanes, int B, int threads, char *text, size_t text_bytes,
                      size_t *text_start);

#ifdef __cplusplus
}
#endif
#endif /* GANET_HIP_H_ */
