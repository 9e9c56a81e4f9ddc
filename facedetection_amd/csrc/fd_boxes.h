// Internal (C++) interface of the box-stage kernels, shared between the C-ABI
// stage entry points and the fused detector pipeline.
#pragma once
#include "fd_common.h"

namespace fd {

// Per-image counters kept on the device: [image][4] int32.
enum { CNT_VALID = 0, CNT_CAND = 1, CNT_KEEP = 2, CNT_STRIDE = 4 };

// Element strides of the RPN head tensors as seen by the proposal decode kernel;
// lets one kernel read the operator layout (NCHW) and the fused NHWC head output.
struct RpnLayout {
    long long cls_batch, cls_anchor, cls_h, cls_w;   // address of the score of (b, a, h, w)
    long long cls_bg_off, cls_fg_off;                // offsets of background / foreground planes
    long long box_batch, box_chan, box_h, box_w;     // address of delta channel c = 4a + k
    long long box_off;
};

// Stable ascending segmented radix sort of (key, value) pairs.  `n` pairs per segment,
// `nseg` segments stored back to back.  Result ends in (keys_a, vals_a).
int sort_pairs_u32(fd_ctx* ctx, uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, int n,
                   int nseg, hipStream_t stream);
int sort_pairs_u64(fd_ctx* ctx, uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, int n,
                   int nseg, hipStream_t stream);

// Greedy NMS over candidates already in visiting order.  cand_boxes [nseg][cap][4];
// counters[seg*CNT_STRIDE + CNT_CAND] = number of candidates of the segment (device).
// keep [nseg][cap] receives candidate ranks; counters[.. + CNT_KEEP] the number kept,
// which stops at max_keep when max_keep > 0.
template <typename T>
int nms_sorted(fd_ctx* ctx, const T* cand_boxes, int32_t* counters, int cap, int nseg, T thresh, int max_keep,
               int32_t* keep, hipStream_t stream);

// Whole RPN proposal stage for `batch` images (see fd_proposal in facedet.h).
// `softmax` != 0: cls holds raw 2-way scores and the pairwise softmax is fused in.
// Image info (h, w, scale): im_info_dev [batch][3] on the device, or, when that is NULL,
// the three host floats info_uniform shared by every image.  rois [batch*post_n][5].
int proposal_run(fd_ctx* ctx, const fd_proposal_cfg* cfg, const float* cls, const float* bbox, const RpnLayout& lay,
                 int softmax, const float* im_info_dev, const float* info_uniform, int batch, int height, int width,
                 float* rois, float* roi_scores, int32_t* counts, int32_t* nan_flag, hipStream_t stream);

// Detector tail for `batch` images of `rois_per_image` rois each (see fd_postprocess).
// rois rows are (x1,y1,x2,y2) when roi_stride == 4 or (b,x1,y1,x2,y2) when 5.
// dets [batch][rois_per_image][5], n_dets [batch].
int postprocess_run(fd_ctx* ctx, const float* rois, int roi_stride, const float* scores, const float* deltas,
                    int batch, int rois_per_image, int im_h, int im_w, float thresh, float nms_thresh, int do_round,
                    float* dets, int32_t* n_dets, hipStream_t stream);

// ROI max pooling, channels-last: data [batch][H][W][C] -> out [num_rois][ph][pw][C].
int roipool_nhwc(const float* data, int batch, int channels, int height, int width, const float* rois, int num_rois,
                 int pooled_h, int pooled_w, float spatial_scale, float* out, hipStream_t stream);
// Pooling written directly as split(relu(max * bn_scale[c] + bn_shift[c])) (channels % 32 == 0).
// `clamped`: device counter of values outside the fp16 range met while splitting, or NULL.
int roipool_nhwc_split(const float* data, int batch, int channels, int height, int width, const float* rois,
                       int num_rois, int pooled_h, int pooled_w, float spatial_scale, const float* bn_scale,
                       const float* bn_shift, void* out_split, unsigned* clamped, hipStream_t stream,
                       bool f32 = false);   // f32: plain float32 activated values instead of the split form

// Resolves, for `device` (the current device), the address of the fixed workspace that
// fd_nms_f32_noalloc uses; called from fd_ctx_create so that the entry point itself never has to.
int noalloc_init_device(int device);

}  // namespace fd
