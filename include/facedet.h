/*
 * facedet.h -- C ABI of the MI355X (gfx950) face-detection hot path.
 *
 * Drop-in boundary for the detector of tanmlh/FaceDetection (code/detection):
 * every entry point below names the reference interface it stands in for.
 * Plain C types only; every pointer is a DEVICE pointer unless its name ends
 * in `_host`.  `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  All calls are asynchronous on `stream` unless stated otherwise and
 * return 0 on success or a negative FD_ERR_* code; fd_last_error() gives the
 * message of the calling thread's last failure.  No exceptions cross the ABI.
 *
 * A `fd_ctx` owns the library's scratch memory for one GPU.  It grows on
 * demand (hipMalloc) and is reused across calls, so steady-state calls do not
 * allocate.  A context must not be used from two threads at once, and calls
 * that share a context must be issued on one stream (the scratch is shared).
 */
#ifndef FACEDET_H
#define FACEDET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FD_OK 0
#define FD_ERR_ARG (-1)    /* bad argument (shape, NULL pointer, unsupported size) */
#define FD_ERR_HIP (-2)    /* a HIP runtime call failed                            */
#define FD_ERR_LIMIT (-3)  /* problem exceeds a documented limit                   */
#define FD_ERR_NAN (-4)    /* NaN found where the reference raises ValueError      */
#define FD_ERR_PARAM (-5)  /* a required weight tensor is missing or mis-shaped    */

typedef struct fd_ctx fd_ctx;
typedef struct fd_detector fd_detector;

/* Library ------------------------------------------------------------------ */
int fd_version(void);
/* Hash (16 hex digits) of the source files this library was built from; profiles record it so that
 * a traffic measurement is only ever quoted for the kernel set it was taken on. */
const char* fd_build_id(void);
const char* fd_last_error(void);
/* One context per GPU per host thread.  Synchronous. */
int fd_ctx_create(int device, fd_ctx** out);
void fd_ctx_destroy(fd_ctx* ctx);
/* Bytes of scratch currently held by the context (diagnostics). */
int64_t fd_ctx_scratch_bytes(const fd_ctx* ctx);

/* Per-kernel timing with HIP events, for benchmarks (no counterpart in the reference, which
 * only prints wall time per frame, detection.py:163-164).  Between fd_profile_begin() and
 * fd_profile_end() every kernel group of the library is bracketed by events on its launch
 * stream.  fd_profile_end synchronises the device and returns one row per kernel group:
 * launches, summed event time, and the summed ALGORITHMIC work of those launches (FLOPs with
 * a multiply-add counted as 2; bytes = each operand tensor read or written once). */
typedef struct fd_profile_row {
    char kernel[64];
    int64_t launches;
    double ms;
    double flops;
    double bytes;
} fd_profile_row;
int fd_profile_begin(void);
/* The same, bracketing only the kernel group called `kernel` (every event costs launch time; a
 * benchmark that needs one kernel's durations inside its timed region asks for that one only). */
int fd_profile_begin_only(const char* kernel);
int fd_profile_end(fd_profile_row* rows_host, int capacity, int* n_rows);

/* Box arithmetic ------------------------------------------------------------ */

/* generate_anchors(base_size, ratios, scales) -- processing.py:184-248.
 * Host-side helper: writes num_ratios*num_scales rows of (x1,y1,x2,y2),
 * ratio-major, into anchors_host (double). Synchronous, no GPU work. */
int fd_generate_anchors(double base_size, const double* ratios_host, int num_ratios,
                        const double* scales_host, int num_scales, double* anchors_host);

/* bbox_pred(boxes, box_deltas) followed (optionally) by clip_boxes(pred, (im_h, im_w))
 * -- processing.py:35-78 and :81-97.  boxes [n,4] float32, deltas [n,4*num_classes]
 * float32, out [n,4*num_classes] float64 (the reference decodes in float64 with exp()
 * taken in float32).  clip = 0 skips the clamp. */
int fd_bbox_pred_f64(fd_ctx* ctx, const float* boxes, const float* deltas, int n, int num_classes,
                     int clip, double im_h, double im_w, double* out, void* stream);

/* clip_boxes(boxes, (im_h, im_w)) -- processing.py:81-97, in place on float64 [n,4*num_classes]. */
int fd_clip_boxes_f64(fd_ctx* ctx, double* boxes, int n, int num_classes, double im_h, double im_w, void* stream);

/* nms(dets, thresh) -- processing.py:114-148.  dets [n,5] (x1,y1,x2,y2,score);
 * keep [n] receives indices into dets in the order boxes were kept (descending
 * score; equal scores: higher index first), *n_keep their number.  IoU arithmetic
 * is done in the dtype of dets, in the reference's operation order, and a box
 * survives a kept one iff iou <= thresh. */
int fd_nms_f32(fd_ctx* ctx, const float* dets, int n, float thresh, int32_t* keep, int32_t* n_keep, void* stream);
int fd_nms_f64(fd_ctx* ctx, const double* dets, int n, double thresh, int32_t* keep, int32_t* n_keep, void* stream);

/* fd_nms_f32 without any allocation.  No counterpart in the reference: it exists for callers that
 * cannot allocate device memory at the point of the call -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  Same arguments, same results and same kernels as
 * fd_nms_f32 for n <= 4096 boxes (FD_ERR_LIMIT beyond); no context: the sort buffers, the
 * suppression mask and the counters live in a fixed 2.3 MB device array of the library, one per
 * GPU, which every GPU the library runs on holds until the process ends.  Operates on the CURRENT
 * device (dets, keep, n_keep and stream must belong to it).  The array's address is resolved by
 * fd_ctx_create for its device, otherwise by the first call on a device (not under graph capture
 * then).  All calls on one GPU share that array: calls that overlap on one GPU (two streams, two
 * host threads) race -- serialise them.  The event profiler must be off (it creates events). */
int fd_nms_f32_noalloc(const float* dets, int n, float thresh, int32_t* keep, int32_t* n_keep, void* stream);

/* RPN proposal operator --------------------------------------------------------
 * mx.symbol.contrib.Proposal(cls_prob, bbox_pred, im_info, ...) as called at
 * symbol/faster_rcnn.py:98-102, computed with the steps of its in-tree Python twin
 * ProposalOperator.forward (symbol/proposal.py:28-142).                          */
typedef struct fd_proposal_cfg {
    int feat_stride;       /* 16                                              */
    int num_scales;        /* <= 8                                            */
    float scales[8];       /* (4, 8, 16, 32)                                  */
    int num_ratios;        /* <= 8                                            */
    float ratios[8];       /* (0.5, 1, 2)                                     */
    int pre_nms_top_n;     /* config.TEST.RPN_PRE_NMS_TOP_N; <= 0 = no cut    */
    int post_nms_top_n;    /* config.TEST.RPN_POST_NMS_TOP_N; must be > 0     */
    float nms_thresh;      /* config.TEST.RPN_NMS_THRESH                      */
    float min_size;        /* config.TEST.RPN_MIN_SIZE                        */
    int box_f64;           /* 1: decode + NMS in float64 (what the Python twin
                              computes); 0: float32 throughout                */
} fd_proposal_cfg;

/* cls_prob [batch, 2A, H, W] float32 (foreground = channels A..2A-1), bbox_pred
 * [batch, 4A, H, W] float32, im_info_host [batch,3] = (height, width, scale) on the
 * HOST.  rois [batch*post_nms_top_n, 5] float32 = (batch index, x1, y1, x2, y2),
 * roi_scores [batch*post_nms_top_n] float32 (may be NULL), counts [batch,3] int32
 * (may be NULL) = (boxes passing the min-size filter, candidates given to NMS,
 * distinct boxes written to rois, i.e. min(kept by NMS, post_nms_top_n)).  A short keep list is
 * padded cyclically; an image with no valid box yields all-zero rois.
 * The reference operator rejects batch > 1 (proposal.py:176-177); this entry point
 * accepts it and treats images independently.  *nan_flag (int32, may be NULL) is
 * set non-zero when a NaN is met in the scores or deltas (proposal.py:47-51). */
int fd_proposal(fd_ctx* ctx, const fd_proposal_cfg* cfg, const float* cls_prob, const float* bbox_pred,
                const float* im_info_host, int batch, int height, int width, float* rois, float* roi_scores,
                int32_t* counts, int32_t* nan_flag, void* stream);

/* ROI max pooling ---------------------------------------------------------------
 * mx.symbol.ROIPooling(data, rois, pooled_size, spatial_scale) as called at
 * symbol/faster_rcnn.py:103.  data [batch, C, H, W] float32, rois [num_rois,5]
 * (batch index, x1, y1, x2, y2), out [num_rois, C, pooled_h, pooled_w]. */
int fd_roipool(fd_ctx* ctx, const float* data, int batch, int channels, int height, int width, const float* rois,
               int num_rois, int pooled_h, int pooled_w, float spatial_scale, float* out, void* stream);

/* Detector tail -------------------------------------------------------------------
 * Detector.detect_face lines detection.py:47-58: decode rois with the head's deltas in
 * float64, clip to (im_h, im_w), take class 1, keep score >= thresh, cast to float32,
 * greedy NMS (nms_thresh), rows in kept order, coordinates rounded half-to-even when
 * do_round != 0.  rois [num_rois,4] float32, scores [num_rois,2], deltas [num_rois,8];
 * dets [num_rois,5] float32 receives *n_dets rows. */
int fd_postprocess(fd_ctx* ctx, const float* rois, const float* scores, const float* deltas, int num_rois,
                   int im_h, int im_w, float thresh, float nms_thresh, int do_round, float* dets, int32_t* n_dets,
                   void* stream);

/* Image resize -----------------------------------------------------------------------
 * cv2.resize(im, None, None, fx, fy, interpolation=cv2.INTER_LINEAR) on 8-bit images, the call
 * inside resize() at detection.py:63-78 (OpenCV's fixed-point bilinear arithmetic; see
 * csrc/fd_image.hip).  src uint8 [batch, src_h, src_w, channels] -> dst uint8
 * [batch, dst_h, dst_w, channels] with (dst_h, dst_w) as given by fd_resize_dims. */
int fd_resize_dims(int src_h, int src_w, double fx, double fy, int* dst_h_host, int* dst_w_host);
int fd_resize_u8(fd_ctx* ctx, const uint8_t* src, int batch, int src_h, int src_w, int channels, double fx,
                 double fy, uint8_t* dst, int dst_h, int dst_w, void* stream);
/* The same with cv2.cvtColor(color, cv2.COLOR_BGR2RGB) (detection.py:109) folded in when
 * reverse_channels != 0: destination channel k holds source channel channels-1-k. */
int fd_resize_cvt_u8(fd_ctx* ctx, const uint8_t* src, int batch, int src_h, int src_w, int channels, double fx,
                     double fy, int reverse_channels, uint8_t* dst, int dst_h, int dst_w, void* stream);

/* Face crops ------------------------------------------------------------------------
 * Function.stick_boxes (code/common/Function.py:49-70) followed by Function.resize_pad
 * (code/common/Function.py:95-113) for every box of a frame, as the reference's video loop
 * does between tracker and recogniser (code/data_processing/ProcessVideo.py:150,
 * Function.py:381-382).  frame uint8 [height, width, channels] on the device; boxes float64
 * [n_boxes, box_stride] on the device, columns 0..3 = x1, y1, x2, y2 (detections or tracker
 * rows).  tiles uint8 [n_boxes, dst_h, dst_w, channels] receives the letterboxed crops;
 * rects int32 [n_boxes, 4] (may be NULL) receives the padded, clamped slice bounds
 * (x1, y1, x2, y2; x2 / y2 exclusive).  The reference draws a rectangle into the frame before
 * slicing; that overlay is not reproduced.  An empty rectangle gives a black tile.  With
 * tiles == NULL only the rectangles are computed. */
int fd_crop_resize_pad(fd_ctx* ctx, const uint8_t* frame, int height, int width, int channels, const double* boxes,
                       int box_stride, int n_boxes, int dst_h, int dst_w, uint8_t* tiles, int32_t* rects,
                       void* stream);

/* Convolution operator ---------------------------------------------------------------
 * mx.sym.Convolution as used throughout symbol/faster_rcnn.py:33-52,85-89,131, with the
 * neighbouring BatchNorm / Activation / elementwise-add operators it is fused with here:
 *   y = [relu]( conv( pre(x) ) + bias ) + residual,   pre(x) = relu(x * pre_scale + pre_shift)
 * Tensors are channels-last: x [batch, H, W, cin], y / residual [batch, Ho, Wo, cout];
 * w_oihw [cout, cin, kh, kw] in the reference's (MXNet) weight layout.  bias [cout],
 * pre_scale / pre_shift [cin] (only for pad == 0) and residual may be NULL.  cin must be
 * a multiple of 32.  Arithmetic: float32 multiply-accumulate on the matrix cores. */
int fd_conv2d(fd_ctx* ctx, const float* x_nhwc, int batch, int height, int width, int cin, const float* w_oihw,
              int cout, int kh, int kw, int stride, int pad, const float* bias, const float* pre_scale,
              const float* pre_shift, int relu, const float* residual, float* y_nhwc, void* stream);

/* The same operator, same float32 tensors, with every product formed on the bf16 matrix cores from three-way bf16
 * splits of both operands (FD_PRECISION_BF16X3, see below); cout must be a multiple of 8, tensors below 2^31 bytes. */
int fd_conv2d_bf16x3(fd_ctx* ctx, const float* x_nhwc, int batch, int height, int width, int cin,
                     const float* w_oihw, int cout, int kh, int kw, int stride, int pad, const float* bias,
                     const float* pre_scale, const float* pre_shift, int relu, const float* residual,
                     float* y_nhwc, void* stream);

/* The same operator with split-precision arithmetic (FD_PRECISION_F16X3, see below); cout must be a
 * multiple of 8.  Reads the weights back to the host to choose their scale: synchronous. */
int fd_conv2d_f16x3(fd_ctx* ctx, const float* x_nhwc, int batch, int height, int width, int cin,
                    const float* w_oihw, int cout, int kh, int kw, int stride, int pad, const float* bias,
                    const float* pre_scale, const float* pre_shift, int relu, const float* residual,
                    float* y_nhwc, void* stream);

/* The 3x3, stride 1, pad 1 case of the same operator in Winograd form, float32 (the form the detector uses
 * for the RPN's 3x3 convolution, faster_rcnn.py:85-87, and the wide 3x3 layers of faster_rcnn.py:41-43 in
 * FD_PRECISION_F32): y = [relu](conv3x3(x) + bias).  tile = 2: F(2x2,3x3), 4/9 of the multiplies of the
 * direct form; tile = 4: F(4x4,3x3), 1/4 of them (interpolation points 0, +-5/8, +-3/2).  cin a multiple of
 * 32, cout of 8.  Transforms the weights on the host: synchronous. */
int fd_conv2d_winograd(fd_ctx* ctx, const float* x_nhwc, int batch, int height, int width, int cin, const float* w_oihw,
                       int cout, const float* bias, int relu, int tile, float* y_nhwc, void* stream);

/* Detector ------------------------------------------------------------------------------
 * Detector.__init__ (detection.py:26-31): takes the checkpoint's tensors by their MXNet names
 * (arg_params and aux_params of `mxnet-face-fr50`: conv0_weight, bn0_gamma, bn0_moving_mean,
 * stage1_unit1_conv1_weight, ..., rpn_conv_3x3_weight/bias, cls_score_weight/bias, ...),
 * HOST float32 arrays in MXNet layout, and builds faster_rcnn50(num_class=2)
 * (symbol/faster_rcnn.py:174-179) on the context's GPU.  The arrays are copied; synchronous.
 * `cfg` carries the config.TEST.RPN_* values that the reference bakes into the graph. */
typedef struct fd_param {
    const char* name;
    const float* data_host;
    int64_t numel;
} fd_param;

int fd_detector_create(fd_ctx* ctx, const fd_param* params, int n_params, const fd_proposal_cfg* cfg,
                       fd_detector** out);
void fd_detector_destroy(fd_detector* det);
/* Arithmetic of the convolutions.  All accumulate in float32 on the matrix cores.  In all, the RPN's 3x3
 * convolution and the wide stride-1 3x3 layers of the bottleneck units are computed in Winograd F(4x4,3x3)
 * form (a quarter of the multiplies; same end-to-end error against float64 as the direct form, see
 * fd_conv2d_winograd); environment variable FD_WINOGRAD=0, read by fd_detector_create, keeps every layer direct.
 *   FD_PRECISION_F32   float32 operands (v_mfma_f32_16x16x4_f32): the reference's arithmetic.  Default.
 *   FD_PRECISION_F16X3 every float32 operand is split exactly into two fp16 numbers
 *                      (hi + lo, ~22 significant bits) and each product is formed as
 *                      hi*hi + hi*lo + lo*hi on the f16 matrix cores: float32-grade results
 *                      (same error level against a float64 reference) at 16/3 of the rate.
 *                      Activations are clamped to the fp16 range (+-65504); every clamped value
 *                      is counted (fd_detector_status).  Opt-in.
 *   FD_PRECISION_BF16X3 the float32 tensors, weights and kernels of FD_PRECISION_F32, but each product is formed on
 *                      the bf16 matrix cores: both operands are cut in registers into three bf16 numbers
 *                      (hi + mid + lo reproduces a float32 value to 2^-27 of its size, with the float32 exponent range:
 *                      nothing is clamped) and a product is the six partial products down to 2^-18 of its size,
 *                      accumulated in float32 -- the dropped ones are below a float32 multiply's own rounding.  The
 *                      bf16 matrix unit rounds its sums downwards; the kernels cancel that bias by accumulating
 *                      the last 37 % of every K loop negated.  Measured against float64 the error equals that of
 *                      FD_PRECISION_F32 (fd_conv2d with FD_BF16X3=1 in the environment: rms error 0.85-1.15 times
 *                      the float32 kernel's on K = 128 ... 4608, mean error below 1e-8 of the output rms), at up to
 *                      8/3 of the fp32 matrix rate.  The fused stem keeps float32 products.  Opt-in. */
#define FD_PRECISION_F32 0
#define FD_PRECISION_F16X3 1
#define FD_PRECISION_BF16X3 2
int fd_detector_set_precision(fd_detector* det, int precision);

/* Replaces the proposal settings (the reference mutates `config` and rebuilds the symbol). */
int fd_detector_set_proposal(fd_detector* det, const fd_proposal_cfg* cfg);

/* exe.forward(is_train=False) (detection.py:42-46): frames uint8 [batch, H, W, 3] (any channel
 * order: the graph does not care) -> the graph's three outputs, all float32 on the device:
 * rois [batch*R, 5] (`rpn_rois_output`), cls_prob [batch*R, 2] (`cls_prob_reshape_output`),
 * bbox_pred [batch*R, 8] (`bbox_pred_reshape_output`), R = post_nms_top_n.  Any of the three
 * may be NULL.  im_scale is im_info[2] (1 in detect_face, the resize factor in the CLI). */
int fd_forward(fd_detector* det, const uint8_t* frames, int batch, int height, int width, float im_scale,
               float* rois, float* cls_prob, float* bbox_pred, void* stream);

/* Detector.detect_face (detection.py:33-60) for `batch` frames of one size: forward pass plus
 * the tail of fd_postprocess.  dets [batch, R, 5] float32, n_dets [batch] int32: image b has
 * n_dets[b] valid rows (x1, y1, x2, y2, score), best score first.  *nan_flag (device int32,
 * may be NULL) is non-zero if the RPN produced a NaN (the reference raises ValueError).
 * With FD_PRECISION_F32 a pass takes as many frames as keep every tensor below 2^31 bytes (16 at
 * 1080p with 10 proposals per frame); a larger batch is run as several passes inside the call
 * (fd_detect_scaled as well), a frame's rows do not depend on it, and the status block covers all
 * passes.  fd_forward has one pass and fails for such a batch. */
int fd_detect(fd_detector* det, const uint8_t* frames, int batch, int height, int width, float thresh,
              float nms_thresh, int do_round, float* dets, int32_t* n_dets, int32_t* nan_flag, void* stream);

/* Copies an intermediate tensor of the LAST forward pass (channels-last float32) into `out`
 * (device, `capacity` floats) and writes its 4 dimensions; out == NULL only queries the
 * dimensions.  Names: conv0, pool0, stage{1..4}_unit{j}, feat, rpn_out, rois, roi_scores,
 * pooled, flat, cls_prob, bbox_pred.  For parity tests and debugging.  Layer buffers are
 * recycled inside a pass, so by default only feat and the tensors after it (not pooled) still hold
 * their values afterwards.  fd_detector_keep_tensors(det, keep) makes later passes snapshot
 * tensors (allocating: debugging only):
 *   keep = 1  every named tensor; the pass then runs layer by layer, because the fused kernels of
 *             FD_PRECISION_F16X3 (stem+pool, whole residual units, pooling into the split form)
 *             never materialise conv0, pool0 or pooled;
 *   keep = 2  every tensor the production kernels materialise (stage{1..4}_unit{j}, feat and what
 *             follows): kernel selection is exactly that of an ordinary pass, unit outputs are
 *             additionally written as float32 where an ordinary pass would skip that;
 *   keep = 0  off. */
/* The command-line loop's variant (detection.py:110-148): frames are already resized by
 * im_scale (im_info = (height, width, im_scale), so the proposal stage's minimum size scales
 * with it), the tail does not round, and each kept box is mapped back to the source frame as
 * int(round(coordinate / im_scale)) -- boxes_src int32 [batch, rois_per_image, 4], may be
 * NULL.  dets float32 [batch, rois_per_image, 5] holds the unrounded rows in resized-frame
 * pixels.  Rounding is half-to-even on the float64 quotient. */
int fd_detect_scaled(fd_detector* det, const uint8_t* frames, int batch, int height, int width, float im_scale,
                     float thresh, float nms_thresh, float* dets, int32_t* n_dets, int32_t* boxes_src,
                     int32_t* nan_flag, void* stream);

/* No counterpart in the reference (its executor launches operator by operator): one detection call
 * recorded as a HIP graph and replayed, for callers that run the same frame shape again and again
 * (a video loop, one frame at a time) and want the host's launch cost and the gaps between the
 * pass's short kernels out of the latency.  fd_detect_graph_create runs the call once in the
 * ordinary way (all buffers reach their size), then records it: fd_detect when im_scale == 1 and
 * boxes_src == NULL, fd_detect_scaled with do_round = 0 otherwise; `status` (device int32 [2],
 * may be NULL) receives the status block of fd_detector_status.  All pointers are fixed by the
 * recording: a replay reads `frames` and writes `dets`, `n_dets`, `boxes_src`, `status` at the
 * addresses given here, with the precision and proposal settings the detector had then.
 * fd_detect_graph_launch replays on `stream`; it fails (and nothing is launched) if the library
 * re-allocated a buffer since the recording (a larger shape ran on the same context): record again.
 * Replays and other calls on the same detector or context must not overlap, as for any two calls. */
typedef struct fd_detect_graph fd_detect_graph;
int fd_detect_graph_create(fd_detector* det, const uint8_t* frames, int batch, int height, int width, float im_scale,
                           float thresh, float nms_thresh, int do_round, float* dets, int32_t* n_dets,
                           int32_t* boxes_src, int32_t* status, fd_detect_graph** out);
int fd_detect_graph_launch(fd_detect_graph* graph, void* stream);
void fd_detect_graph_destroy(fd_detect_graph* graph);

/* Status of the LAST forward pass of this detector, copied to status [2] (device int32) on `stream`
 * (the stream the pass was issued on):
 *   status[0]  non-zero if the proposal stage met a NaN in its scores or deltas -- where the reference
 *              raises ValueError (symbol/proposal.py:47-51);
 *   status[1]  number of values that lay outside the fp16 range (|v| > 65504, Inf, NaN) when an
 *              operand was split for FD_PRECISION_F16X3 and were clamped (saturating at 2^31 - 1 is
 *              not attempted: a count that large wraps).  Always 0 for FD_PRECISION_F32.  A non-zero
 *              count means the pass did not have float32-grade accuracy: repeat it with
 *              FD_PRECISION_F32 (the Python Detector does).  The reference has no counterpart: it
 *              computes in unbounded float32 throughout.
 * Parameters themselves are checked when the detector is created: a NaN or Inf in any tensor makes
 * fd_detector_create fail with FD_ERR_PARAM. */
int fd_detector_status(fd_detector* det, int32_t* status, void* stream);

int fd_detector_keep_tensors(fd_detector* det, int keep);
int fd_detector_tensor(fd_detector* det, const char* name, float* out, int64_t capacity, int64_t* dims4,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FACEDET_H */
