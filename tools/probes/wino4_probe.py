"""Error and speed of the two Winograd forms: operator level and detector level (FD_WINOGRAD=0/2/4)."""
import os, sys, json, subprocess
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "det":
    from facedetection_amd.detection import Detector
    from facedetection_amd.synthetic import synthetic_image, synthetic_params
    from oracle import network as onet
    from oracle import boxes as ob
    params = synthetic_params(0)
    PROP = dict(pre_nms_top_n=1000, post_nms_top_n=10, nms_thresh=0.7, min_size=50)
    det = Detector(params, 0, precision="f32")
    worst = [0, 0]
    for seed in range(int(sys.argv[2])):
        img = synthetic_image(1080, 1920, seed=500 + seed)
        got = det.detect_face(img)
        rois, cls_prob, deltas, _ = onet.forward(img, params, PROP, torch.float64)
        ref = ob.postprocess(rois[:, 1:], cls_prob, deltas, (1080, 1920), do_round=False) if 'do_round' in ob.postprocess.__code__.co_varnames else None
        dets, n, st = det.detect_batch_device(det._frames(img), do_round=False)
        g = dets[0, :int(n[0])].cpu().numpy()
        if ref is not None and ref.shape == g.shape:
            worst[0] = max(worst[0], float(np.abs(g[:, 4] - ref[:, 4]).max(initial=0)))
            worst[1] = max(worst[1], float(np.abs(g[:, :4] - ref[:, :4]).max(initial=0)))
        else:
            print("shape mismatch", seed, None if ref is None else ref.shape, g.shape)
    print("FD_WINOGRAD", os.environ.get("FD_WINOGRAD"), "score err %.3g box err %.3g px" % tuple(worst))
    det.close()
else:
    from facedetection_amd import ops
    rng = np.random.default_rng(0)
    for (b, h, w, cin, cout) in [(2, 68, 120, 1024, 512), (2, 68, 120, 256, 256), (40, 7, 7, 512, 512)]:
        x = np.maximum(rng.standard_normal((b, h, w, cin)), 0).astype(np.float32)
        wt = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
        ref = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt).double(), padding=1).permute(0, 2, 3, 1).numpy()
        rms = np.sqrt((ref ** 2).mean())
        out = {}
        for prec in ("f32", "winograd", "winograd4"):
            got = ops.conv2d(x, wt, stride=1, pad=1, precision=prec)
            out[prec] = ("rms %.3g max %.3g" % (np.sqrt(((got - ref) ** 2).mean()) / rms, np.abs(got - ref).max() / np.abs(ref).max()))
        print((b, h, w, cin, cout), out)
