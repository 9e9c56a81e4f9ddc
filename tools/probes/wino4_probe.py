#!/usr/bin/env python3
"""Error of the float32 3x3 convolution forms against float64.

    python tools/probes/wino4_probe.py            operator level: direct, F(2x2,3x3), F(4x4,3x3) on three layer shapes
    FD_WINOGRAD=0|2|4 python tools/probes/wino4_probe.py det [frames=12]
                                                  detector level: unrounded detections of 1080p frames in float32
                                                  against the float64 oracle (worst score and box error)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def detector_level(n_frames):
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image, synthetic_params
    from oracle import boxes as ob
    from oracle import network as onet
    params = synthetic_params(0)
    prop = dict(pre_nms_top_n=1000, post_nms_top_n=10, nms_thresh=0.7, min_size=50)
    det = Detector(params, 0, precision="f32")
    worst = [0.0, 0.0]
    for seed in range(n_frames):
        img = synthetic_image(1080, 1920, seed=500 + seed)
        rois, cls_prob, deltas, _ = onet.forward(img, params, prop, torch.float64)
        ref = ob.postprocess(rois[:, 1:], cls_prob, deltas, (1080, 1920), do_round=False)
        dets, n, _ = det.detect_batch_device(det._frames(img), do_round=False)
        got = dets[0, :int(n[0])].cpu().numpy()
        if ref.shape != got.shape:
            print("frame %d: %d detections, the oracle has %d" % (seed, got.shape[0], ref.shape[0]))
            continue
        worst[0] = max(worst[0], float(np.abs(got[:, 4] - ref[:, 4]).max(initial=0)))
        worst[1] = max(worst[1], float(np.abs(got[:, :4] - ref[:, :4]).max(initial=0)))
    det.close()
    print("FD_WINOGRAD=%s over %d frames: score error %.3g, box error %.3g px"
          % (os.environ.get("FD_WINOGRAD", "(default)"), n_frames, worst[0], worst[1]))


def operator_level():
    from facedetection_amd import ops
    rng = np.random.default_rng(0)
    for (b, h, w, cin, cout) in [(2, 68, 120, 1024, 512), (2, 68, 120, 256, 256), (40, 7, 7, 512, 512)]:
        x = np.maximum(rng.standard_normal((b, h, w, cin)), 0).astype(np.float32)      # as after a ReLU
        wt = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
        ref = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt).double(),
                                         padding=1).permute(0, 2, 3, 1).numpy()
        rms = np.sqrt((ref ** 2).mean())
        out = {}
        for form in ("f32", "winograd", "winograd4"):
            got = ops.conv2d(x, wt, stride=1, pad=1, precision=form)
            out[form] = "rms %.3g max %.3g" % (np.sqrt(((got - ref) ** 2).mean()) / rms,
                                               np.abs(got - ref).max() / np.abs(ref).max())
        print((b, h, w, cin, cout), out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "det":
        detector_level(int(sys.argv[2]) if len(sys.argv) > 2 else 12)
    else:
        operator_level()
