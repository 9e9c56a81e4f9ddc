"""Throughput mode: micro-batches of a frame batch spread over several HIP streams.

The network alternates between kernels that wait on HBM (the 1x1 convolutions of the first
stages) and kernels that keep the matrix cores busy (the 3x3 convolutions).  One stream runs
them one after the other; with two micro-batches in flight on two streams the GPU co-schedules
workgroups of both kinds and each hides part of the other's idle resource (measured: +6 %
frames/s at 1080p, micro-batches of 8).  Every stream has its own detector, i.e. its own
scratch context and activation buffers; the weights are uploaded once per detector.

Results are identical to ``Detector.detect_batch``: frames never interact.
"""
import torch

from .detection import Detector


class FramePipeline:
    def __init__(self, model_path, ctx=None, streams=2, **detector_kwargs):
        if streams < 1:
            raise ValueError("streams must be >= 1")
        self.detectors = [Detector(model_path, ctx, own_context=(streams > 1), **detector_kwargs)
                          for _ in range(streams)]
        self.device = self.detectors[0].device
        with torch.cuda.device(self.device):
            self.streams = [torch.cuda.Stream() for _ in range(streams)] if streams > 1 else [None]

    def close(self):
        for d in self.detectors:
            d.close()

    def reconfigure(self):
        for d in self.detectors:
            d.reconfigure()

    def set_precision(self, precision):
        for d in self.detectors:
            d.set_precision(precision)

    def detect_device(self, frames, micro_batch=8, nms_thresh=0.3, thresh=0.8, do_round=True, streams=None):
        """frames uint8 [N,H,W,3] on the GPU -> list, in frame order, of
        (dets [b,R,5], n_dets [b], status [2] = (NaN flag, clamped values)) device tensors per
        micro-batch.  Asynchronous:
        the caller's current stream waits for all side streams before this returns, so work
        queued on it afterwards (or a synchronize) sees the results.  ``streams`` limits how many
        of the pipeline's streams take part (1: everything on the caller's current stream)."""
        outs = []
        use = self.streams if streams is None else self.streams[:max(1, streams)]
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            if use[0] is None or len(use) == 1:
                for i in range(0, frames.shape[0], micro_batch):
                    outs.append(self.detectors[0].detect_batch_device(frames[i:i + micro_batch], nms_thresh, thresh,
                                                                      do_round))
                return outs
            for s in use:
                s.wait_stream(cur)
            for j, i in enumerate(range(0, frames.shape[0], micro_batch)):
                k = j % len(use)
                with torch.cuda.stream(use[k]):
                    outs.append(self.detectors[k].detect_batch_device(frames[i:i + micro_batch], nms_thresh, thresh,
                                                                      do_round))
            for s in use:
                cur.wait_stream(s)
        return outs

    def detect(self, frames, micro_batch=8, nms_thresh=0.3, thresh=0.8):
        """List of float32 [n_i,5] arrays, one per frame (host)."""
        f = self.detectors[0]._frames(frames)
        outs = self.detect_device(f, micro_batch, nms_thresh, thresh, True)
        res = []
        for j, (dets, n, status) in enumerate(outs):
            det = self.detectors[j % len(self.detectors)]
            if det._check_status(status):
                # values left the fp16 range: this micro-batch again with float32 operands
                res += det.detect_batch(f[j * micro_batch:(j + 1) * micro_batch], nms_thresh, thresh, True)
                continue
            d, c = dets.cpu().numpy(), n.cpu().numpy()
            res += [d[b, :c[b]].copy() for b in range(d.shape[0])]
        return res
