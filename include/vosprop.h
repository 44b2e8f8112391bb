/*
 * vosprop - MI355X-native label-propagation engine (C ABI, the drop-in boundary).
 *
 * This header is the whole boundary of the hot path.  The reference project
 * (hynekdav/semi-supervised-VOS) has no FFI layer; the hot path is three Python call sites.
 * Each entry point below names the reference interface (file:line in the reference tree) it
 * replaces.  Plain C types only: pointers + sizes, no torch / HIP types (streams are void*).
 *
 * Conventions
 *   - every function returns 0 on success or a negative VOSPROP_E_* code; nothing throws.
 *   - `vosprop_last_error(ctx)` gives a human-readable message for the last failure on ctx.
 *   - pointers named *_dev are device (HBM) pointers on the ctx's GPU; *_host are host pointers.
 *   - the engine BORROWS input buffers for the duration of the call and copies what it keeps
 *     into its own ring; outputs are written to caller-owned device buffers.
 *   - one ctx per (GPU, stream); a ctx is not thread-safe; there is no global state.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     no call synchronises the device except vosprop_begin_video / vosprop_begin_video_labels (use their
 *     *_on forms to stay stream-ordered) and vosprop_destroy.
 */
#ifndef VOSPROP_H
#define VOSPROP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOSPROP_ABI_VERSION 1

/* error codes */
#define VOSPROP_OK 0
#define VOSPROP_E_INVALID (-1)      /* bad argument / configuration                       */
#define VOSPROP_E_HIP (-2)          /* a HIP runtime call or kernel launch failed         */
#define VOSPROP_E_STATE (-3)        /* call order violated (e.g. step before begin_video) */
#define VOSPROP_E_UNSUPPORTED (-4)  /* valid in the reference but outside this engine     */
#define VOSPROP_E_NOMEM (-5)

/* element types of feature tensors handed to the engine */
#define VOSPROP_DT_F32 0
#define VOSPROP_DT_F16 1
#define VOSPROP_DT_BF16 2
/* OR into a feature dtype: the tensor is channels-last, (H_d*W_d, C) pixel-major - what a channels_last encoder hands over, and the
 * engine's own ring order (a bf16 push is then a plain copy).  Without it feature tensors are (C, H_d, W_d). */
#define VOSPROP_LAYOUT_HWC 0x10

/* arithmetic of the affinity contraction */
#define VOSPROP_PREC_BF16 0 /* bf16 MFMA (v_mfma_f32_32x32x16_bf16), f32 accumulate - the fast path */
#define VOSPROP_PREC_F32 1  /* f32-input MFMA (v_mfma_f32_32x32x2_f32), exact f32 - the parity path  */

/* limits of this build */
#define VOSPROP_MAX_CLASSES 32 /* d = objects + 1 handled per propagation pass */
#define VOSPROP_MAX_REF 64     /* sampled reference frames per step (ref_num)  */
#define VOSPROP_MAX_DIM 256    /* H_d, W_d <= 256 (coordinates exact in bf16)  */

typedef struct vosprop_ctx vosprop_ctx;

/* Mirrors the `main.py inference` options that reach the hot path
 * (reference src/inference.py:19-31,42-43) plus the engine's own knobs. */
typedef struct vosprop_config {
    int abi_version;      /* must be VOSPROP_ABI_VERSION                                        */
    int device;           /* HIP device ordinal                                                  */
    int feat_h, feat_w;   /* H_d, W_d = ceil(H/8), ceil(W/8)  (reference src/model/predict.py:109-110) */
    int channels;         /* C; this build requires 256 (reference src/model/vos_net.py:22-23)   */
    int ref_num;          /* --ref_num      (src/inference.py:19), default 9                     */
    int frame_range;      /* --frame_range  (src/inference.py:27), default 40                    */
    float sigma1;         /* --sigma_1      (src/inference.py:28), default 8                     */
    float sigma2;         /* --sigma_2      (src/inference.py:30), default 21                    */
    float temperature;    /* --temperature  (src/inference.py:26), default 1; must be > 0        */
    int probability;      /* --probability  (src/inference.py:42): 0 = propagate one-hot labels  */
    int topk;             /* 0 = dense (the reference); 1..32 = keep the k largest A[.,t] per target pixel, zero the rest,
                             no renormalisation (NOT in the reference; label-propagation mode only; one scoring pass + a re-score
                             of the reference tiles that can hold a kept entry; at most 32 768 reference tiles per step) */
    int precision;        /* VOSPROP_PREC_*                                                      */
    int ring_capacity;    /* 0 = auto: max(frame_range + 4, ref_num) + 1 frames; anything smaller is VOSPROP_E_INVALID */
    int materialise;      /* 0 = fused (nothing of size (H_d W_d)^2 touches HBM).  1 = the reference's algorithm SHAPE
                             (src/model/predict.py:49-55): the (N HW) x HW affinity is written to HBM as bf16 and read back -
                             2 x N HW^2 x 2 bytes per step (7.46 GB at 720p), an HBM-bandwidth-bound variant kept for measurement
                             (BASELINE.json configs[4]); dense, bf16 path only.  (Was reserved[0]: old callers pass 0.)          */
    int reserved[7];      /* zero                                                                */
} vosprop_config;

/* Fill cfg with the reference CLI defaults for a (feat_h, feat_w) map. */
void vosprop_default_config(vosprop_config* cfg, int feat_h, int feat_w);

/* Library / build information, e.g. "vosprop 0.1 gfx950".  Never NULL. */
const char* vosprop_version(void);

/* Create / destroy an engine context (allocates the feature+label ring in HBM). */
int vosprop_create(vosprop_ctx** out, const vosprop_config* cfg);
void vosprop_destroy(vosprop_ctx* ctx);
const char* vosprop_last_error(const vosprop_ctx* ctx);

/* Start a video.  Replaces `prepare_first_frame` (reference src/model/predict.py:99-155, 'single'
 * branch) minus the PNG I/O: first_label_host is the decoded 00000.png, (H,W) uint8 class indices.
 * Computes d = max+1 (predict.py:113), the nearest-down-sampled one-hot labels (predict.py:92-96),
 * resets the ring.  The spatial weights (predict.py:117-118, 158-175) are never materialised: the
 * kernel evaluates them from pixel coordinates.  *d_out receives d.  Requires
 * ceil(H/8)==feat_h, ceil(W/8)==feat_w and d <= VOSPROP_MAX_CLASSES. */
int vosprop_begin_video(vosprop_ctx* ctx, const uint8_t* first_label_host, int H, int W, int* d_out);

/* Same, for the strategies whose feature map is not ceil(H/8) x ceil(W/8) of the output image (reference
 * prepare_first_frame branches '2-scale' / 'hor-2-scale' / '3-scale', src/model/predict.py:137-153) or whose first labels are
 * transformed (flips, :130-135): the caller passes the already down-sampled class map (feat_h x feat_w, uint8), the class
 * count d and the size (out_h, out_w) the masks of vosprop_step are up-sampled to. */
int vosprop_begin_video_labels(vosprop_ctx* ctx, const uint8_t* cls_lowres_host, int d, int out_h, int out_w);

/* Stream-ordered forms of the two calls above: the label upload, the label packing and the state reset are enqueued on `stream`
 * (the stream the vosprop_step calls of this context use) and the call returns without waiting for the device, so a loop that
 * walks many videos never drains the GPU at a video boundary.  first_label_host / cls_lowres_host are consumed before return. */
int vosprop_begin_video_on(vosprop_ctx* ctx, const uint8_t* first_label_host, int H, int W, int* d_out, void* stream);
int vosprop_begin_video_labels_on(vosprop_ctx* ctx, const uint8_t* cls_lowres_host, int d, int out_h, int out_w,
                                  void* stream);

/* One iteration of the `inference_single` loop body after the encoder
 * (reference src/utils/inference_utils.py:33-75):
 *   frame 0  : stores the features in the ring (:36, feats_history = model(input)); no outputs.
 *   frame i>0: prediction = predict(...) (:56-65, src/model/predict.py:19-71); new label =
 *              one-hot(argmax) or the prediction itself in probability mode (:67-70); history append
 *              (:71-72); nearest up-sample + argmax (:74-75).
 * feat_dev   (C, H_d, W_d) in NCHW order - or (H_d*W_d, C) with VOSPROP_LAYOUT_HWC - element type feat_dtype (the encoder output
 *            features[0]).  Stream-ordered like every device argument: it must stay unchanged until the work this call enqueues
 *            on `stream` has run (channels-last bf16 features are read in place by the propagation kernel and copied into the
 *            ring by the kernel after it, not up front).
 * pred_out_dev  optional (d, H_d*W_d) f32 - the reference's `prediction` tensor.  NULL in label mode lets the step skip the softmax
 *            denominators (they do not change the arg-max): same mask, same new label.
 * mask_out_dev  optional (H, W) uint8    - the reference's per-frame mask (class indices).
 * The frame index is kept by the engine (0,1,2,... since begin_video). */
int vosprop_step(vosprop_ctx* ctx, const void* feat_dev, int feat_dtype, float* pred_out_dev,
                 uint8_t* mask_out_dev, void* stream);

/* Frame index the NEXT vosprop_step call will process (0 right after begin_video). */
int vosprop_frame_index(const vosprop_ctx* ctx);

/* Stateless operator: the reference's
 *   predict(ref, target, ref_label, weight_dense, weight_sparse, frame_idx, range, ref_num,
 *           temperature, probability_propagation)            (src/model/predict.py:19-28)
 * with the two (HW,HW) weight matrices replaced by their sigmas.
 *   ref_dev       (T, C, H_d, W_d)   history features, element type feat_dtype, T >= frame_idx
 *   target_dev    (C, H_d, W_d)
 *   ref_label_dev (d, T, H_d*W_d)    f32 (one-hot or probabilities)
 *   out_dev       (d, H_d*W_d)       f32
 * Uses ctx's GPU, precision, top-k and a scratch ring of its own (sized by the number of frames the sampler returns).
 * Does not touch the video state of ctx. */
int vosprop_predict(vosprop_ctx* ctx, const void* ref_dev, const void* target_dev, int feat_dtype,
                    const float* ref_label_dev, int T, int d, int frame_idx, int frame_range, int ref_num,
                    float temperature, float sigma1, float sigma2, int probability, float* out_dev,
                    void* stream);

/* Encoder epilogue, context-free: y = act(y + bias[c] (+ residual)) in place over a channels-last tensor viewed as
 * (pixels, channels); replaces the bias / BatchNorm shift, the `out += identity` and the ReLU that follow every convolution of the
 * reference's residual units (src/model/backbone/resnet.py:45-58, 81-95) by one pass.  channels % 8 == 0; y, bias, residual share
 * `dtype` (VOSPROP_DT_*); residual may be NULL; relu != 0 applies max(., 0).  Device pointers, enqueued on `stream`. */
int vosprop_bias_act(void* y, const void* bias, const void* residual, long long pixels, int channels, int relu, int dtype,
                     void* stream);

/* Stem epilogue, context-free: y[n, oh, ow, c] = relu(max_{3x3 window, stride 2, pad 1} x[n, ., ., c] + bias[c]) - the BatchNorm
 * shift, ReLU and 3x3/2 max-pool after the 7x7 stem convolution (src/model/backbone/resnet.py:104-107) in one pass over the
 * channels-last activation x (n, h, w, channels); y is (n, (h-1)/2+1, (w-1)/2+1, channels).  Bit-identical to bias + ReLU (rounded
 * to `dtype`) followed by the pool, because all three are monotone.  channels % 8 == 0; x, bias, y share `dtype`. */
int vosprop_bias_relu_maxpool(const void* x, const void* bias, void* y, int n, int h, int w, int channels, int dtype,
                              void* stream);

/* Pointwise (1x1, stride 1, no padding) convolution with its epilogue, context-free:
 *     y[p, co] = act( sum_ci x[p, ci] * weight[co, ci] + bias[co] (+ residual[p, co]) )
 * over channels-last tensors viewed as (pixels, channels) - conv1 / conv3 / downsample[0] of the reference's bottleneck units and
 * `adjust_dim` together with the BatchNorm shift, `out += identity` and ReLU that follow them (src/model/backbone/resnet.py:66-95,
 * src/model/vos_net.py:27-52) as ONE hipBLASLt GEMM (f32 accumulation) whose epilogue does the rest, so the output is written
 * once.  x (pixels, cin), weight (cout, cin), bias (cout), residual / y (pixels, cout) share `dtype`; bias and residual may be
 * NULL; residual may alias y.  The first call for a problem (pixels, cin, cout, epilogue) - never inside a stream capture - computes
 * an f32 reference of up to 1 024 sampled output rows, runs the library's candidate algorithms on the operands, and lets only those
 * whose output agrees with the reference within the rounding of the output type compete on time (a fast candidate with wrong or
 * sloppy numerics can never win); the winner gets a workspace of its own.  Which algorithm won is remembered by its library
 * solution index in $VOSPROP_CACHE_DIR (default ~/.cache/vosprop; VOSPROP_PW_CACHE=0: off); a later process re-validates the
 * remembered algorithm against the reference before trusting it.  An algorithm is never reused for another problem size.  Keep the
 * calls for one device stream-ordered (one stream at a time, as the encoder wrapper does).  VOSPROP_E_UNSUPPORTED: the library has
 * no validated kernel for the shape, or the first call for the problem came inside a stream capture - use a convolution +
 * vosprop_bias_act. */
int vosprop_pointwise_conv(const void* x, const void* weight, const void* bias, const void* residual, void* y,
                           long long pixels, int cin, int cout, int relu, int dtype, void* stream);

/* Reproducible mode, process-wide (also switched on by VOSPROP_DETERMINISTIC=1 in the environment): vosprop_pointwise_conv then
 * picks, for every problem, the FIRST candidate in the library's own rank order that passes the numerical gate - no timing race
 * between candidates, no cache file read or written - so the kernel of a layer is a pure function of the problem and two processes
 * (a one-process run and the shards of `main.py inference --gpus N --deterministic`) compute the same bits.  The propagation kernels
 * themselves are deterministic in every mode (fixed work map, partials merged in a fixed order, no atomics on the dense path).
 * The reference has no such switch: its videos are independent (src/utils/inference_utils.py:28-48), which is what makes "sharded
 * output == single-process output" a byte-level statement.  Returns the previous setting. */
int vosprop_set_deterministic(int on);

/* Frame sampler, reference `sample_frames` (src/model/predict.py:74-89).  Host-side, exact.
 * out must hold max(num_refs, 3) ints (or frame_idx when frame_idx <= num_refs); returns the count - for num_refs == 3 past
 * frame 3 that is the 3 "continuous" frames alone.  num_refs < 3 past frame num_refs: VOSPROP_E_INVALID (the reference's
 * np.linspace is asked for a negative number of samples and raises ValueError). */
int vosprop_sample_frames(int frame_idx, int frame_range, int num_refs, int* out);

/* Timing / introspection of the last propagation on ctx (for bench.py and the roofline line). */
typedef struct vosprop_stats {
    int n_ref;              /* sampled reference frames N                          */
    int hw;                 /* H_d*W_d                                             */
    int workgroups;         /* grid of the propagation kernel                      */
    int tiles_per_wg;       /* reference tiles (32 rows) each workgroup walks      */
    double flops;           /* algorithmic FLOP of the step, counted once: 2*N*HW^2*C + 2*d*N*HW^2 (top-k: + 2*d*k*HW) */
    double bytes;           /* algorithmic bytes: N*HW*C*2 + HW*C*2 + N*HW + d*HW*4   */
    int kernel_id;          /* which propagation kernel the engine LAUNCHED for the step: VOSPROP_KERNEL_* (set where the launch is
                               decided, so a bench line or a test never re-derives the dispatch rule) */
    int reserved_;
} vosprop_stats;
/* vosprop_stats.kernel_id */
#define VOSPROP_KERNEL_DENSE        1   /* prop_dense_kernel: dense, softmax denominators kept (a prediction was requested, or probability mode) */
#define VOSPROP_KERNEL_MASK         2   /* prop_mask_kernel: the hand-ordered mask-only label-mode step (frame loop, bench.py) */
#define VOSPROP_KERNEL_TOPK         3   /* prop_dense_kernel<TK 1> + topk_select2 + prop_dense_kernel<TK 2> */
#define VOSPROP_KERNEL_F32          4   /* prop_f32_kernel (VOSPROP_PREC_F32) */
#define VOSPROP_KERNEL_MATERIALISED 5   /* prop_dense_kernel<MAT 1> + <MAT 2> (affinity through HBM) */
/* Top-k variant: the three capacity limits of its kernels are REPORTED, never silent.  out3 = how often, since the context's first
 * top-k step, (0) a lane ran out of dump slots for a candidate group in the re-scoring pass, (1) a pixel had more than 40 candidate
 * groups in the combine kernel, (2) a pixel had more than 512 candidate list entries in the select kernel - each drops candidates
 * (the result is then a lower bound of the exact top-k sum).  All zero on every shape of the test-suite and the bench except the
 * constructed mass-tie case.  Waits for the work enqueued on `stream`. */
int vosprop_topk_overflows(vosprop_ctx* ctx, unsigned* out3, void* stream);

/* name of a VOSPROP_KERNEL_* value ("prop_mask_kernel" ...), or "?" */
const char* vosprop_kernel_name(int kernel_id);
int vosprop_last_stats(const vosprop_ctx* ctx, vosprop_stats* out);

/* Measure the propagation kernel alone: re-runs the last step's propagation `iters` times on
 * `stream` between HIP events recorded on that stream and returns the mean kernel time in
 * microseconds (bench.py's roofline.achieved).  Outputs are rewritten identically. */
int vosprop_time_last_propagation(vosprop_ctx* ctx, int iters, void* stream, double* mean_us);

/* In-situ timing of the propagation kernel: between vosprop_timing_begin and vosprop_timing_read every dense propagation launch of
 * vosprop_step on ctx (label mode) carries a pair of HIP events attached to the dispatch itself on the launch stream
 * (hipExtLaunchKernelGGL: the kernel's own begin / end timestamps, no queue latency; at most 4096 launches are kept).  vosprop_timing_read waits for the recorded launches and returns their mean duration in microseconds and their
 * number.  This is the kernel as it runs inside the loop (cold caches, neighbours on the stream) - what bench.py's
 * roofline.achieved is computed from; vosprop_time_last_propagation gives the back-to-back figure. */
int vosprop_timing_begin(vosprop_ctx* ctx);
int vosprop_timing_read(vosprop_ctx* ctx, double* mean_us, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* VOSPROP_H */
