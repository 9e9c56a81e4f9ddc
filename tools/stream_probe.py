#!/usr/bin/env python3
"""Throughput of the 1080p pipeline with micro-batches spread over S HIP streams (one detector
with its own scratch context per stream), to see how much the HBM-bound and the MFMA-bound
kernels of different micro-batches overlap.

    python tools/stream_probe.py [frames=128] [micro_batch=8] [streams=1,2,3] [cu_split]

With a fourth argument the S streams are created with complementary CU masks (stream k gets every
S-th CU of each XCD), so kernels of different micro-batches run side by side on disjoint CUs
instead of one after the other.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_frames  # noqa: E402
from facedetection_amd.detection import Detector  # noqa: E402


def masked_stream(k, S, n_cu=256):
    """A HIP stream limited to the CUs with index k modulo S, wrapped for torch."""
    import ctypes
    # the HIP runtime this process already uses (torch ships its own copy)
    path = "libamdhip64.so"
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
    hip = ctypes.CDLL(path)
    words = n_cu // 32
    mask = (ctypes.c_uint32 * words)()
    # The runtime only accepts masks that are whole bytes: with 8 XCDs a byte is one CU index on
    # every XCD, so byte j -> stream j modulo S gives each stream the same share of every XCD.
    for j in range(n_cu // 8):
        if j % S == k:
            mask[j // 4] |= 0xFF << (8 * (j % 4))
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return torch.cuda.ExternalStream(h.value)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ss = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    frames = make_frames(n, dev, seed=0)
    for S in ss:
        dets = [Detector("synthetic:0", 0, own_context=True, precision="f16x3") for _ in range(S)]
        if len(sys.argv) > 4 and S > 1:
            streams = [masked_stream(k, S) for k in range(S)]
        else:
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
