#!/usr/bin/env python3
"""Throughput of the 1080p pipeline with micro-batches spread over S HIP streams (one detector
with its own scratch context per stream), to see how much the HBM-bound and the MFMA-bound
kernels of different micro-batches overlap.

    python tools/stream_probe.py [frames=128] [micro_batch=8] [streams=1,2,3]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_frames  # noqa: E402
from facedetection_amd.detection import Detector  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ss = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    frames = make_frames(n, dev, seed=0)
    for S in ss:
        dets = [Detector("synthetic:0", 0, own_context=True) for _ in range(S)]
        streams = [torch.cuda.Stream() for _ in range(S)]

        def step():
            outs = []
            for j, i in enumerate(range(0, n, B)):
                k = j % S
                with torch.cuda.stream(streams[k]):
                    outs.append(dets[k].detect_batch_device(frames[i:i + B]))
            return outs

        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            outs = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tot = int(sum(int(o[1].sum()) for o in outs))
        print("streams %d  micro-batch %d: %.1f frames/s  (%d detections in last step)" % (S, B, 3 * n / dt, tot), flush=True)
        for d in dets:
            d.close()


if __name__ == "__main__":
    main()
