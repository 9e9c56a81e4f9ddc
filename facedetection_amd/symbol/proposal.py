"""Proposal operator with the CustomOp-style protocol of the reference's
code/detection/symbol/proposal.py (class and method names, argument/output lists,
shape inference and its errors), executing on the GPU through ``ops.proposal``.

There is no MXNet here, so ``in_data`` / ``out_data`` are plain arrays: ``forward``
reads ``in_data = [cls_prob, bbox_pred, im_info]`` and stores its results into the
list ``out_data`` (index 0 rois, index 1 scores when ``output_score``)."""
import numpy as np

from .. import ops
from .config import config
from .processing import generate_anchors


def _parse(seq):
    if isinstance(seq, str):            # MXNet passes operator attributes as strings: "(4, 8, 16, 32)"
        return [float(v) for v in seq.strip("()[] ").split(",") if v.strip()]
    return [float(v) for v in seq]


class ProposalOperator(object):
    def __init__(self, feat_stride, scales, ratios, is_train=False, output_score=False):
        self._feat_stride = float(feat_stride)
        self._scales = _parse(scales)
        self._ratios = _parse(ratios)
        self._anchors = generate_anchors(base_size=self._feat_stride, scales=self._scales, ratios=self._ratios)
        self._num_anchors = self._anchors.shape[0]
        self._output_score = output_score
        if is_train:
            raise NotImplementedError("only the TEST configuration is on the detection path")
        self.cfg_key = "TEST"

    def forward(self, is_train, req, in_data, out_data, aux=None):
        cfg = config[self.cfg_key]
        cls_prob, bbox_pred, im_info = in_data[0], in_data[1], in_data[2]
        res = ops.proposal(cls_prob, bbox_pred, im_info, output_score=self._output_score,
                           feature_stride=int(self._feat_stride), scales=self._scales, ratios=self._ratios,
                           rpn_pre_nms_top_n=cfg.RPN_PRE_NMS_TOP_N, rpn_post_nms_top_n=cfg.RPN_POST_NMS_TOP_N,
                           threshold=cfg.RPN_NMS_THRESH, rpn_min_size=cfg.RPN_MIN_SIZE)
        if self._output_score:
            self.assign(out_data, 0, res[0])
            self.assign(out_data, 1, res[1])
        else:
            self.assign(out_data, 0, res)

    @staticmethod
    def assign(out_data, index, value):
        while len(out_data) <= index:
            out_data.append(None)
        out_data[index] = value

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux=None):
        for i in (0, 1):                # the operator has no gradient
            self.assign(in_grad, i, 0)


class ProposalProp(object):
    def __init__(self, feat_stride, scales, ratios, is_train=False, output_score=False):
        self._feat_stride = feat_stride
        self._scales = scales
        self._ratios = ratios
        self._is_train = is_train in (True, "True")
        self._output_score = output_score
        self.cfg_key = "TRAIN" if self._is_train else "TEST"

    def list_arguments(self):
        return ["cls_prob", "bbox_pred", "im_info"]

    def list_outputs(self):
        return ["output", "score"] if self._output_score else ["output"]

    def infer_shape(self, in_shape):
        cls_prob_shape, bbox_pred_shape = in_shape[0], in_shape[1]
        assert cls_prob_shape[0] == bbox_pred_shape[0], "ROI number does not equal in cls and reg"
        batch_size = cls_prob_shape[0]
        if batch_size > 1:
            raise ValueError("Only single item batches are supported")
        post_n = config[self.cfg_key].RPN_POST_NMS_TOP_N
        outs = [(post_n, 5)]
        if self._output_score:
            outs.append((post_n, 1))
        return [cls_prob_shape, bbox_pred_shape, (batch_size, 3)], outs

    def create_operator(self, ctx=None, shapes=None, dtypes=None):
        return ProposalOperator(self._feat_stride, self._scales, self._ratios, self._is_train, self._output_score)

    def declare_backward_dependency(self, out_grad, in_data, out_data):
        return []
