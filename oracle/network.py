"""torch-CPU oracle of the detector's graph (test infrastructure).

Restates the MXNet symbol built by the reference:
  code/detection/symbol/faster_rcnn.py:6-53     residual_unit (pre-activation bottleneck)
  code/detection/symbol/faster_rcnn.py:73-104   rpn head, Proposal, ROIPooling
  code/detection/symbol/faster_rcnn.py:107-162  resnet: stem, 4 stages, head
  code/detection/symbol/faster_rcnn.py:174-179  faster_rcnn50 = units [3,4,6,3], filters [64,256,512,1024,2048]
  code/common/Function.py:82-87                 np2nd: HWC -> 1x3xHxW float32, no mean subtraction
  code/detection/detection.py:33-60             Detector.detect_face

Operators follow MXNet 1.x inference semantics (MXNet itself is neither in the reference
tree nor installed; its version is not pinned by the reference):
  BatchNorm  y = (x - moving_mean) / sqrt(moving_var + eps) * gamma + beta, eps = 2e-5,
             gamma forced to 1 when fix_gamma=True (bn_data);
  Convolution / FullyConnected as in torch; Pooling max 3x3/2 pad 1 with floor ("valid")
  output size; global average pooling; SoftmaxActivation over the channel axis.

PARITY UNPINNED for the network forward: the reference ships no weights, no fixtures and
no runnable MXNet, so nothing but the graph definition pins this file.  The box stages it
calls (oracle.boxes, oracle.roipool) carry their own parity notes.

``dtype=torch.float64`` runs the whole graph in double precision and is the yardstick for
the 1e-4 score tolerance; ``torch.float32`` is the like-for-like CPU baseline.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import boxes as ob
from . import roipool as orp

EPS = 2e-5
UNITS = (3, 4, 6, 3)
FILTERS = (64, 256, 512, 1024, 2048)


def _t(params, name, dtype):
    return torch.from_numpy(np.asarray(params[name])).to(dtype)


def batch_norm(x, params, name, dtype, fix_gamma=False):
    mean = _t(params, name + "_moving_mean", dtype)
    var = _t(params, name + "_moving_var", dtype)
    beta = _t(params, name + "_beta", dtype)
    gamma = torch.ones_like(beta) if fix_gamma else _t(params, name + "_gamma", dtype)
    return F.batch_norm(x, mean, var, gamma, beta, training=False, eps=EPS)


def conv(x, params, name, dtype, stride=1, pad=0, bias=False):
    w = _t(params, name + "_weight", dtype)
    b = _t(params, name + "_bias", dtype) if bias else None
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def residual_unit(x, params, name, dtype, stride, dim_match):
    """faster_rcnn.py:27-53 (bottleneck branch)."""
    act1 = F.relu(batch_norm(x, params, name + "_bn1", dtype))
    c1 = conv(act1, params, name + "_conv1", dtype)
    act2 = F.relu(batch_norm(c1, params, name + "_bn2", dtype))
    c2 = conv(act2, params, name + "_conv2", dtype, stride=stride, pad=1)
    act3 = F.relu(batch_norm(c2, params, name + "_bn3", dtype))
    c3 = conv(act3, params, name + "_conv3", dtype)
    shortcut = x if dim_match else conv(act1, params, name + "_sc", dtype, stride=stride)
    return c3 + shortcut


def stage(x, params, s, dtype):
    """Stage s (1-based): first unit changes width (and halves resolution for s > 1)."""
    x = residual_unit(x, params, "stage%d_unit1" % s, dtype, 1 if s == 1 else 2, False)
    for j in range(2, UNITS[s - 1] + 1):
        x = residual_unit(x, params, "stage%d_unit%d" % (s, j), dtype, 1, True)
    return x


def backbone(data, params, dtype=torch.float32):
    """data [B,3,H,W] -> stage-3 feature map [B,1024,H/16,W/16] (faster_rcnn.py:128-144)."""
    x = batch_norm(data.to(dtype), params, "bn_data", dtype, fix_gamma=True)
    x = conv(x, params, "conv0", dtype, stride=2, pad=3)
    x = F.relu(batch_norm(x, params, "bn0", dtype))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for s in (1, 2, 3):
        x = stage(x, params, s, dtype)
    return x


def rpn_head(feat, params, dtype=torch.float32):
    """faster_rcnn.py:85-94: returns cls_prob [B,24,H,W] (after the 2-way softmax) and bbox_pred [B,48,H,W]."""
    r = F.relu(conv(feat, params, "rpn_conv_3x3", dtype, pad=1, bias=True))
    cls_score = conv(r, params, "rpn_cls_score", dtype, bias=True)
    bbox_pred = conv(r, params, "rpn_bbox_pred", dtype, bias=True)
    B, C2, H, W = cls_score.shape
    prob = F.softmax(cls_score.reshape(B, 2, (C2 // 2) * H, W), dim=1).reshape(B, C2, H, W)
    return prob, bbox_pred, cls_score


def head(pooled, params, dtype=torch.float32):
    """Stage 4 + classifier on pooled rois [R,1024,7,7] (faster_rcnn.py:137-160)."""
    x = stage(pooled.to(dtype), params, 4, dtype)
    x = F.relu(batch_norm(x, params, "bn1", dtype))
    flat = x.mean(dim=(2, 3))
    cls_score = F.linear(flat, _t(params, "cls_score_weight", dtype), _t(params, "cls_score_bias", dtype))
    cls_prob = F.softmax(cls_score, dim=1)
    bbox = F.linear(flat, _t(params, "bbox_pred_weight", dtype), _t(params, "bbox_pred_bias", dtype))
    return cls_prob, bbox


def np2nd(img_np):
    """Function.py:82-87: HWC image -> [1,3,H,W] float32."""
    return torch.from_numpy(np.ascontiguousarray(img_np)).to(torch.float32).permute(2, 0, 1).unsqueeze(0)


def forward(img_np, params, cfg, dtype=torch.float32, im_scale=1.0):
    """Graph outputs for one image: (rois [post_n,5], cls_prob [post_n,2], bbox_pred [post_n,8]) as
    float32 NumPy arrays plus a dict of intermediates.  ``cfg`` has the proposal settings
    (pre_nms_top_n, post_nms_top_n, nms_thresh, min_size, box_dtype)."""
    with torch.no_grad():
        data = np2nd(img_np)
        H, W = data.shape[2], data.shape[3]
        feat = backbone(data, params, dtype)
        prob, bbox_pred, cls_score = rpn_head(feat, params, dtype)
        prob32 = prob.to(torch.float32).numpy()
        bbox32 = bbox_pred.to(torch.float32).numpy()
        rois, roi_scores = ob.proposal(prob32, bbox32, np.array([H, W, im_scale], np.float32),
                                       pre_nms_top_n=cfg["pre_nms_top_n"], post_nms_top_n=cfg["post_nms_top_n"],
                                       nms_thresh=cfg["nms_thresh"], min_size=cfg["min_size"],
                                       dtype=cfg.get("box_dtype", np.float64))
        feat32 = feat.to(torch.float32).numpy()
        pooled = orp.roi_pooling(feat32, rois, (7, 7), 0.0625)
        cls_prob, deltas = head(torch.from_numpy(pooled), params, dtype)
    inter = dict(feat=feat32, rpn_cls_prob=prob32, rpn_bbox_pred=bbox32, rpn_cls_score=cls_score.to(torch.float32).numpy(),
                 roi_scores=roi_scores, pooled=pooled)
    return rois, cls_prob.to(torch.float32).numpy(), deltas.to(torch.float32).numpy(), inter


def detect_face(img_np, params, cfg, nms_thresh=0.3, thresh=0.8, dtype=torch.float32, return_all=False):
    """Detector.detect_face (detection.py:33-60): image uint8/float [H,W,3] -> dets float32 [n,5]."""
    rois, cls_prob, deltas, inter = forward(img_np, params, cfg, dtype)
    dets = ob.postprocess(rois[:, 1:], cls_prob, deltas, img_np.shape[:2], thresh=thresh, nms_thresh=nms_thresh)
    if return_all:
        inter.update(rois=rois, cls_prob=cls_prob, bbox_deltas=deltas)
        return dets, inter
    return dets
