#!/usr/bin/env python3
"""Rates of the kernels either side of the detector (the rows that follow the hot path): the
command-line front end (colour swap + resize of decoded frames, detection on the resized frames)
and the crop hand-off (stick_boxes + resize_pad tiles for the recogniser), on device-resident
frames.  Algorithmic bytes: source pixels read once + destination pixels written once.

    python tools/frontend_bench.py [frames=32]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facedetection_amd import ops  # noqa: E402
from facedetection_amd.detection import Detector  # noqa: E402
from facedetection_amd.synthetic import synthetic_image  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    torch.cuda.set_device(0)
    frames = torch.from_numpy(np.stack([synthetic_image(1080, 1920, seed=s) for s in range(n)])).cuda()
    out = {}
    # resize as the command-line loop does it (short side 600, long side capped at 1000)
    scale = 1000.0 / 1920.0
    t = timed(lambda: ops.resize_bilinear(frames, scale, scale), 10)
    small = ops.resize_bilinear(frames, scale, scale)
    moved = frames.numel() + small.numel()
    # the kernel's own duration (HIP events around the launch): the host loop above also pays for
    # the allocation of the output and the launch itself, a good part of a ~50 us kernel
    from facedetection_amd import _lib, profiling
    profiling.begin()
    for _ in range(20):
        ops.resize_bilinear(frames, scale, scale)
    row = [r for r in profiling.end() if r["kernel"] == "resize_linear_u8"][0]
    tk = row["ms"] * 1e-3 / row["launches"]
    out["resize_1080p_to_%dx%d" % (small.shape[1], small.shape[2])] = {
        "frames": n, "frames_per_s": round(n / t, 1), "GB_per_s": round(moved / t / 1e9, 1),
        "kernel_us": round(tk * 1e6, 2), "kernel_GB_per_s": round(moved / tk / 1e9, 1),
        "kernel_frac_of_8TBs": round(moved / tk / 8e12, 3), "library_build": _lib.lib().fd_build_id().decode()}
    # what plain streaming reaches on this GPU with the same number of bytes (torch's device-to-device copy:
    # half read, half written), as a yardstick next to the 8 TB/s of the data sheet; up to 32 frames the
    # working set fits the 256 MB memory-side cache and both numbers are above what HBM delivers
    half = torch.empty(moved // 2, dtype=torch.uint8, device="cuda")
    half2 = torch.empty_like(half)
    half2.copy_(half)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        half2.copy_(half)
    e1.record()
    torch.cuda.synchronize()
    tc = e0.elapsed_time(e1) * 1e-3 / 20
    out["streaming_yardstick"] = {"copy_bytes": int(2 * half.numel()), "copy_GB_per_s": round(2 * half.numel() / tc / 1e9, 1)}
    key = "resize_1080p_to_%dx%d" % (small.shape[1], small.shape[2])
    out[key]["kernel_frac_of_copy_rate"] = round(out[key]["kernel_GB_per_s"] / out["streaming_yardstick"]["copy_GB_per_s"], 3)
    del half, half2
    # 4K frames through the same rule (scale 1000 / 3840)
    n4 = max(1, n // 4)
    f4 = torch.from_numpy(np.stack([np.tile(synthetic_image(1080, 1920, seed=100 + s), (2, 2, 1)) for s in range(n4)])).cuda()
    s4 = 1000.0 / 3840.0
    small4 = ops.resize_bilinear(f4, s4, s4)
    profiling.begin()
    for _ in range(20):
        ops.resize_bilinear(f4, s4, s4)
    row = [r for r in profiling.end() if r["kernel"] == "resize_linear_u8"][0]
    tk = row["ms"] * 1e-3 / row["launches"]
    moved4 = f4.numel() + small4.numel()
    out["resize_4k_to_%dx%d" % (small4.shape[1], small4.shape[2])] = {
        "frames": n4, "kernel_us": round(tk * 1e6, 2), "kernel_GB_per_s": round(moved4 / tk / 1e9, 1),
        "kernel_GB_per_s_rows_touched": round((f4.numel() * (2.0 * s4) + small4.numel()) / tk / 1e9, 1),
        "note": "kernel_GB_per_s counts the whole source + destination; at this scale only 2 source rows in 3.84 "
                "are tap rows, and only those are fetched (rows_touched counts them alone)"}
    del f4, small4
    # crop hand-off: 16 boxes per frame -> 112x112 tiles
    rng = np.random.default_rng(0)
    boxes = np.zeros((16, 5), dtype=np.float64)
    boxes[:, 0] = rng.uniform(0, 1500, 16); boxes[:, 1] = rng.uniform(0, 800, 16)
    boxes[:, 2] = boxes[:, 0] + rng.uniform(60, 400, 16); boxes[:, 3] = boxes[:, 1] + rng.uniform(60, 270, 16)
    one = frames[0]
    t = timed(lambda: ops.crop_resize_pad(one, boxes, (112, 112)), 50)
    src = float(np.sum((boxes[:, 2] - boxes[:, 0]) * 1.2 * (boxes[:, 3] - boxes[:, 1]) * 1.2 * 3))
    out["crop_16_boxes_to_112x112"] = {"us_per_frame": round(t * 1e6, 1),
                                       "GB_per_s": round((src + 16 * 112 * 112 * 3) / t / 1e9, 2)}
    # the command-line loop end to end on decoded 1080p frames (resize + detect + map back)
    det = Detector("synthetic:0", 0, precision="f16x3")
    t = timed(lambda: det.detect_scaled_batch(frames[:8], thresh=0.5), 5)
    out["cli_loop_1080p_batch8"] = {"frames_per_s": round(8 / t, 1)}
    det.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
