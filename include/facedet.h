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
const char* fd_last_error(void);
/* One context per GPU per host thread.  Synchronous. */
int fd_ctx_create(int device, fd_ctx** out);
void fd_ctx_destroy(fd_ctx* ctx);
/* Bytes of scratch currently held by the context (diagnostics). */
int64_t fd_ctx_scratch_bytes(const fd_ctx* ctx);

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

#ifdef __cplusplus
}
#endif
#endif /* FACEDET_H */
