#!/usr/bin/env python3
"""Latency and per-kernel time of the BASELINE.json configurations other than the headline one:
single 1080p frame, single 4K frame, dense-crowd 1080p frame.  Run on the GPU box."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facedetection_amd import profiling  # noqa: E402
from facedetection_amd.detection import Detector  # noqa: E402
from facedetection_amd.symbol.config import config  # noqa: E402
from facedetection_amd.synthetic import crowd_image, synthetic_image  # noqa: E402


def run(det, frame, reps):
    f = det._frames(frame)
    for _ in range(2):
        det.detect_batch_device(f)
    torch.cuda.synchronize()
    profiling.begin()
    t0 = time.perf_counter()
    for _ in range(reps):
        det.detect_batch_device(f)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    rows = profiling.end()
    top = sorted(rows, key=lambda r: -r["ms"])
    table = [{"kernel": r["kernel"], "launches": r["launches"] // reps, "ms": round(r["ms"] / reps, 3),
              "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1) if r["flops"] else None,
              "gbs": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)} for r in top]
    return dt * 1e3, table


def run_plain(det, frame, reps):
    """The same loop without the event brackets of the kernel table, and as a recorded pass (HIP graph)."""
    f = det._frames(frame)
    for _ in range(2):
        det.detect_batch_device(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        det.detect_batch_device(f)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / reps * 1e3
    rec = det.record(f.shape[0], f.shape[1], f.shape[2])
    rec.frames.copy_(f)
    for _ in range(2):
        rec.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        rec.launch()
    torch.cuda.synchronize()
    recorded = (time.perf_counter() - t0) / reps * 1e3
    rec.close()
    return round(plain, 3), round(recorded, 3)


def run_host(det, frame, reps):
    """detect_face on a host (NumPy) frame: upload over PCIe, detect, read the rows back."""
    for _ in range(2):
        det.detect_face(frame)
    t0 = time.perf_counter()
    for _ in range(reps):
        det.detect_face(frame)
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    det = Detector("synthetic:0", 0, precision=precision)
    out = {}
    out["1080p_single"] = run(det, synthetic_image(1080, 1920, seed=1), 20)
    extra = {"1080p_single": run_plain(det, synthetic_image(1080, 1920, seed=1), 20)}
    print("1080p_single_from_host  %8.2f ms/frame (PCIe upload + detect + read-back)"
          % run_host(det, synthetic_image(1080, 1920, seed=1), 20))
    out["4k_single"] = run(det, np.tile(synthetic_image(1080, 1920, seed=2), (2, 2, 1)), 10)
    extra["4k_single"] = run_plain(det, np.tile(synthetic_image(1080, 1920, seed=2), (2, 2, 1)), 10)
    config.TEST.RPN_MIN_SIZE, config.TEST.RPN_PRE_NMS_TOP_N, config.TEST.RPN_POST_NMS_TOP_N = 12, 0, 6000
    det.reconfigure()
    out["dense_crowd_1080p"] = run(det, crowd_image(1080, 1920, seed=2), 5)
    extra["dense_crowd_1080p"] = run_plain(det, crowd_image(1080, 1920, seed=2), 5)
    for k, (ms, top) in out.items():
        print("%-20s %8.2f ms/frame  %s" % (k, ms, [(r["kernel"], r["ms"]) for r in top[:8]]))
    for k, (plain, recorded) in extra.items():
        print("%-20s %8.3f ms/frame without event brackets, %8.3f ms as a recorded pass" % (k, plain, recorded))
    print(json.dumps({"precision": precision,
                      **{k: {"ms": round(v[0], 3), "ms_without_event_brackets": extra[k][0], "ms_recorded_pass": extra[k][1],
                             "kernels": v[1]} for k, v in out.items()}}))


if __name__ == "__main__":
    main()
