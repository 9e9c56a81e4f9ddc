"""Detector API with the call signature of the reference's code/detection/detection.py.

    det = Detector(model_path, ctx)             # detection.py:26
    dets = det.detect_face(frame)               # detection.py:33 -> float32 [N,5]

``dets`` rows are (x1, y1, x2, y2, score) in input-image pixels, best score first,
coordinates rounded to integers -- what the reference hands to its SORT tracker
(code/tracking/sort.py:185) and to ``Function.stick_boxes`` unchanged.

All arithmetic runs in the HIP library through the C ABI; torch tensors are only
device buffers.  ``model_path`` may be
  * a dict ``name -> float32 array`` with MXNet parameter names (arg + aux params),
  * the path of an ``.npz`` file holding such a dict,
  * an MXNet checkpoint prefix (``prefix-%04d.params``), read by ``checkpoint.load_params``,
  * ``"synthetic"`` / ``"synthetic:<seed>"`` for the seeded random-weight model used in
    tests and benchmarks (the reference's real weights are not redistributable here).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib, ops, runtime
from .symbol.config import config


def _device_of(ctx):
    """GPU index from an int, a torch device / string, or an MXNet-style context object."""
    if ctx is None:
        return runtime.device_index(None)
    if hasattr(ctx, "device_id"):
        return int(ctx.device_id)
    return runtime.device_index(ctx)


def load_model(model_path, epoch=0):
    if isinstance(model_path, dict):
        return model_path
    if isinstance(model_path, str) and model_path.startswith("synthetic"):
        from .synthetic import synthetic_params
        seed = int(model_path.split(":", 1)[1]) if ":" in model_path else 0
        return synthetic_params(seed)
    if isinstance(model_path, str) and model_path.endswith(".npz"):
        with np.load(model_path) as z:
            return {k: z[k] for k in z.files}
    from .checkpoint import load_params
    return load_params("%s-%04d.params" % (model_path, epoch))


class Detector:
    """Faster R-CNN face detector (ResNet-50 backbone) on one MI355X."""

    def __init__(self, model_path, ctx=None, epoch=0, box_f64=True, precision="f32", own_context=False,
                 on_clamp="fallback"):
        """``own_context``: give this detector a library context (scratch memory) of its own
        instead of the per-GPU shared one, so that it can run concurrently with other
        detectors of the process on a different HIP stream.

        ``on_clamp``: what the host-side calls (``detect_face``, ``detect_batch``, ...) do when a
        pass in "f16x3" precision had to clamp a value to the fp16 range, i.e. left the range in
        which that arithmetic is as good as the reference's float32: "fallback" (default) repeats
        the batch in "f32" precision, "raise" raises OverflowError, "ignore" returns the clamped
        result.  ``clamp_fallbacks`` counts the repeated batches."""
        if on_clamp not in ("fallback", "raise", "ignore"):
            raise ValueError("on_clamp must be 'fallback', 'raise' or 'ignore'")
        self.on_clamp = on_clamp
        self.clamp_fallbacks = 0
        config.TEST.RPN_MIN_SIZE = 50            # detection.py:27
        self.device = _device_of(ctx)
        self.ctx = ctx
        self._box_f64 = box_f64
        params = load_model(model_path, epoch)
        self._lib = _lib.lib()
        with torch.cuda.device(self.device):
            self._rt = runtime.Context(self.device) if own_context else runtime.context(self.device)
            self._own_rt = own_context
            # the parameter arrays must stay alive and contiguous during the call only
            keep = []
            arr = (_lib.Param * len(params))()
            for i, (name, value) in enumerate(params.items()):
                a = np.ascontiguousarray(np.asarray(value), dtype=np.float32)
                keep.append(a)
                arr[i].name = name.encode("ascii")
                arr[i].data_host = a.ctypes.data
                arr[i].numel = a.size
            self._cfg = self._proposal_cfg()
            handle = C.c_void_p()
            _lib.check(self._lib.fd_detector_create(self._rt.handle, arr, len(params), C.byref(self._cfg),
                                                    C.byref(handle)))
            self._h = handle
            self.set_precision(precision)

    def set_precision(self, precision):
        """"f32" (default): float32 operands on the fp32 matrix cores, the reference's arithmetic.
        "f16x3": exact two-way fp16 split of both operands, three f16 MFMAs per product, float32
        accumulate -- the same error level against float64 inside the fp16 range, about twice the
        frame rate; values that leave the range are counted and handled as ``on_clamp`` says.
        "bf16x3": the float32 tensors and kernels with every product formed on the bf16 matrix cores from exact
        three-way bf16 splits of both operands (six partial products, float32 accumulate, float32 range, nothing
        clamped): the error of "f32" against float64 at a higher rate."""
        code = {"f32": 0, "f16x3": 1, "bf16x3": 2}[precision]
        _lib.check(self._lib.fd_detector_set_precision(self._h, code))
        self.precision = precision

    def _drop_recorded(self):
        """Recorded passes bake in the proposal settings of the moment they were made."""
        for rec in self.__dict__.pop("_recorded", {}).values():
            rec.close()

    def _proposal_cfg(self):
        # values are read from the global config when the detector is built or re-configured,
        # like the reference bakes them into its symbol (symbol/faster_rcnn.py:98-102)
        return ops.make_proposal_cfg(feature_stride=16, scales=(4, 8, 16, 32), ratios=(0.5, 1, 2),
                                     box_f64=self._box_f64)

    def reconfigure(self):
        """Re-read config.TEST.RPN_* (after the caller changed them)."""
        self._cfg = self._proposal_cfg()
        _lib.check(self._lib.fd_detector_set_proposal(self._h, C.byref(self._cfg)))
        self._drop_recorded()

    def close(self):
        self._drop_recorded()
        if getattr(self, "_h", None):
            self._lib.fd_detector_destroy(self._h)
            self._h = None
            if getattr(self, "_own_rt", False):
                self._rt.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _frames(self, img):
        """uint8 device tensor [B,H,W,3] from one HWC image or a batch of them."""
        if isinstance(img, torch.Tensor):
            t = img
            if t.dtype != torch.uint8:
                raise TypeError("frames must be uint8")
            if t.dim() == 3:
                t = t.unsqueeze(0)
            return t.to("cuda:%d" % self.device).contiguous()
        a = np.asarray(img)
        if a.ndim == 3:
            a = a[None]
        if a.ndim != 4 or a.shape[3] != 3:
            raise ValueError("expected an image [H,W,3] or a batch [B,H,W,3]")
        if a.dtype != np.uint8:
            b = a.astype(np.uint8)
            if not np.array_equal(b, a):
                raise TypeError("frames must hold uint8 pixel values")
            a = b
        return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:%d" % self.device)

    @property
    def rois_per_image(self):
        return int(self._cfg.post_nms_top_n)

    # ------------------------------------------------------------------ API
    def forward(self, img, im_scale=1.0):
        """The graph's three outputs (detection.py:44-46) as NumPy arrays:
        rois [B*R,5], cls_prob [B*R,2], bbox_pred [B*R,8]."""
        with torch.cuda.device(self.device):
            f = self._frames(img)
            B, H, W, _ = f.shape
            R = self.rois_per_image
            rois = runtime.empty((B * R, 5), torch.float32, self.device)
            cls = runtime.empty((B * R, 2), torch.float32, self.device)
            box = runtime.empty((B * R, 8), torch.float32, self.device)
            _lib.check(self._lib.fd_forward(self._h, runtime.ptr(f), B, H, W, float(im_scale), runtime.ptr(rois),
                                            runtime.ptr(cls), runtime.ptr(box), runtime.stream_ptr(self.device)))
            if self._check_status(self.status_device()):
                return self._in_f32(lambda: self.forward(f, im_scale))
            return rois.cpu().numpy(), cls.cpu().numpy(), box.cpu().numpy()

    def status_device(self):
        """Status block of the last pass as a device int32 [2] tensor: (NaN flag, clamped values)."""
        status = runtime.empty((2,), torch.int32, self.device)
        _lib.check(self._lib.fd_detector_status(self._h, runtime.ptr(status), runtime.stream_ptr(self.device)))
        return status

    def clamped_values(self):
        """Values the last pass had to clamp to the fp16 range (host read; 0 in "f32" precision)."""
        with torch.cuda.device(self.device):
            return int(self.status_device()[1].item())

    def detect_batch_device(self, frames, nms_thresh=0.3, thresh=0.8, do_round=True):
        """Asynchronous batched detection: frames uint8 [B,H,W,3] already on the GPU.
        Returns device tensors (dets [B,R,5], n_dets [B], status [2] = (NaN flag, clamped values));
        no host sync, and therefore no float32 fallback: the caller inspects ``status``."""
        B, H, W, _ = frames.shape
        R = self.rois_per_image
        dets = runtime.empty((B, R, 5), torch.float32, self.device)
        n = runtime.empty((B,), torch.int32, self.device)
        status = runtime.empty((2,), torch.int32, self.device)
        stream = runtime.stream_ptr(self.device)
        _lib.check(self._lib.fd_detect(self._h, runtime.ptr(frames), B, H, W, float(thresh), float(nms_thresh),
                                       1 if do_round else 0, runtime.ptr(dets), runtime.ptr(n), None, stream))
        _lib.check(self._lib.fd_detector_status(self._h, runtime.ptr(status), stream))
        return dets, n, status

    def _check_status(self, status):
        """Host read of a pass's status block: raises like the reference on NaN; returns True when the
        caller has to repeat the pass in float32 (values were clamped and ``on_clamp`` says so)."""
        nan, clamped = (int(v) for v in status.cpu().numpy())
        if nan != 0:
            raise ValueError("there is nan in input scores or bbox_deltas")
        if clamped != 0 and self.precision == "f16x3":
            if self.on_clamp == "raise":
                raise OverflowError("%d values left the fp16 range of the f16x3 arithmetic" % clamped)
            return self.on_clamp == "fallback"
        return False

    def _in_f32(self, fn):
        """Runs fn() with the detector switched to float32 operands, then switches back."""
        self.clamp_fallbacks += 1
        self.set_precision("f32")
        try:
            return fn()
        finally:
            self.set_precision("f16x3")

    def detect_batch(self, frames, nms_thresh=0.3, thresh=0.8, do_round=True):
        """List of float32 [n_i,5] arrays, one per frame of a same-size batch."""
        with torch.cuda.device(self.device):
            f = self._frames(frames)
            dets, n, status = self.detect_batch_device(f, nms_thresh, thresh, do_round)
            if self._check_status(status):
                return self._in_f32(lambda: self.detect_batch(f, nms_thresh, thresh, do_round))
            n_host = n.cpu().numpy()
            dets_host = dets.cpu().numpy()
            return [dets_host[b, :n_host[b]].copy() for b in range(f.shape[0])]

    def detect_scaled_batch(self, frames, target_size=600, max_size=1000, nms_thresh=0.3, thresh=0.8, bgr=True):
        """The command-line loop of detection.py:105-148 for a same-size batch of decoded frames
        (uint8 [B,H,W,3], BGR as a video decoder delivers them when ``bgr``): colour order
        swapped to RGB, short side resized to ``target_size`` (long side capped at ``max_size``),
        forward with im_info = (h, w, scale), tail without rounding, boxes mapped back with
        int(round(x / scale)).  Returns (list of float32 [n_i,5] rows in resized-frame pixels,
        list of int32 [n_i,4] boxes in source-frame pixels, scale)."""
        with torch.cuda.device(self.device):
            f = self._frames(frames)
            B, H, W, _ = (int(v) for v in f.shape)
            scale = float(target_size) / float(min(H, W))
            if np.round(scale * max(H, W)) > max_size:
                scale = float(max_size) / float(max(H, W))
            dh, dw = C.c_int32(0), C.c_int32(0)
            _lib.check(self._lib.fd_resize_dims(H, W, scale, scale, C.byref(dh), C.byref(dw)))
            small = runtime.empty((B, dh.value, dw.value, 3), torch.uint8, self.device)
            stream = runtime.stream_ptr(self.device)
            _lib.check(self._lib.fd_resize_cvt_u8(self._rt.handle, runtime.ptr(f), B, H, W, 3, scale, scale,
                                                  1 if bgr else 0, runtime.ptr(small), dh.value, dw.value, stream))
            R = self.rois_per_image
            dets = runtime.empty((B, R, 5), torch.float32, self.device)
            n = runtime.empty((B,), torch.int32, self.device)
            src = runtime.empty((B, R, 4), torch.int32, self.device)
            status = runtime.empty((2,), torch.int32, self.device)
            _lib.check(self._lib.fd_detect_scaled(self._h, runtime.ptr(small), B, dh.value, dw.value, scale,
                                                  float(thresh), float(nms_thresh), runtime.ptr(dets), runtime.ptr(n),
                                                  runtime.ptr(src), None, stream))
            _lib.check(self._lib.fd_detector_status(self._h, runtime.ptr(status), stream))
            if self._check_status(status):
                return self._in_f32(lambda: self.detect_scaled_batch(f, target_size, max_size, nms_thresh, thresh, bgr))
            n_host = n.cpu().numpy()
            d_host, s_host = dets.cpu().numpy(), src.cpu().numpy()
            return ([d_host[b, :n_host[b]].copy() for b in range(B)],
                    [s_host[b, :n_host[b]].copy() for b in range(B)], scale)

    def detect_face(self, img_np, nms_thresh=0.3, thresh=0.8):
        """detection.py:33-60: one HWC uint8 frame -> float32 [N,5]."""
        return self.detect_batch(img_np, nms_thresh, thresh, True)[0]

    def record(self, batch, height, width, nms_thresh=0.3, thresh=0.8, do_round=True):
        """The ``detect_batch`` call for frames of this shape recorded as a HIP graph: a ``RecordedDetect``
        whose ``launch`` replays the whole pass with one launch (a single 1080p frame: about 70 kernel
        launches less on the host and no gaps between the short kernels).  It keeps the precision and
        proposal settings the detector has now."""
        return RecordedDetect(self, batch, height, width, nms_thresh, thresh, do_round)

    def detect_face_replayed(self, img_np, nms_thresh=0.3, thresh=0.8):
        """``detect_face`` through a recorded pass (made on first use for each frame size and setting;
        same results).  For loops over frames of one size, e.g. a video."""
        a = np.asarray(img_np)
        cfg = self._cfg
        key = (a.shape[0], a.shape[1], float(nms_thresh), float(thresh), self.precision,
               int(cfg.pre_nms_top_n), int(cfg.post_nms_top_n), float(cfg.min_size), float(cfg.nms_thresh))
        cache = self.__dict__.setdefault("_recorded", {})
        for attempt in (0, 1):
            rec = cache.get(key)
            if rec is None or rec.stale:
                if rec is not None:
                    rec.close()
                rec = cache[key] = self.record(1, a.shape[0], a.shape[1], nms_thresh, thresh, True)
            try:
                return rec.detect(a[None])[0]
            except _lib.FaceDetError:
                # a larger shape ran since the recording and moved the library's buffers: record again, once
                if attempt or not rec.stale:
                    raise

    def detect(self, img_np, nms_thresh=0.3, thresh=0.8):
        """(boxes_and_scores [N,5], landmarks [N,0]).  The reference model predicts no
        landmarks, so the second array is always empty; it is there for callers that
        expect a boxes/scores/landmarks triple."""
        dets = self.detect_face(img_np, nms_thresh, thresh)
        return dets, np.zeros((dets.shape[0], 0), dtype=np.float32)

    def keep_tensors(self, keep=True, production=False):
        """Debugging: make later forward passes snapshot intermediate tensors.  By default every
        named tensor, which runs the pass layer by layer; with ``production`` the kernels of an
        ordinary pass run and the tensors they materialise (unit outputs, feat, ...) are kept."""
        _lib.check(self._lib.fd_detector_keep_tensors(self._h, (2 if production else 1) if keep else 0))

    def tensor(self, name):
        """An intermediate tensor of the last forward pass as a NumPy array (channels-last)."""
        with torch.cuda.device(self.device):
            dims = (C.c_int64 * 4)()
            _lib.check(self._lib.fd_detector_tensor(self._h, name.encode("ascii"), None, 0, dims, None))
            shape = tuple(int(v) for v in dims)
            out = runtime.empty(shape, torch.float32, self.device)
            _lib.check(self._lib.fd_detector_tensor(self._h, name.encode("ascii"), runtime.ptr(out), out.numel(),
                                                    dims, runtime.stream_ptr(self.device)))
            return out.cpu().numpy()


class RecordedDetect:
    """One ``fd_detect`` call recorded as a HIP graph (``fd_detect_graph_create``) with fixed device buffers:
    ``frames`` uint8 [B,H,W,3] is the input the replay reads, ``dets`` [B,R,5], ``n`` [B] and ``status`` [2]
    are what it writes.  ``launch`` replays on the current stream (asynchronous); ``detect`` is the host-level
    call with the status check of ``Detector.detect_batch``.  No counterpart in the reference."""

    def __init__(self, det, batch, height, width, nms_thresh, thresh, do_round):
        self._det, self._lib, self.device = det, det._lib, det.device
        self._args = (float(nms_thresh), float(thresh), bool(do_round))
        with torch.cuda.device(self.device):
            R = det.rois_per_image
            self.frames = torch.zeros((batch, height, width, 3), dtype=torch.uint8, device="cuda:%d" % self.device)
            self.dets = torch.zeros((batch, R, 5), dtype=torch.float32, device=self.frames.device)
            self.n = torch.zeros((batch,), dtype=torch.int32, device=self.frames.device)
            self.status = torch.zeros((2,), dtype=torch.int32, device=self.frames.device)
            torch.cuda.current_stream().synchronize()          # the recording runs on a stream of its own
            handle = C.c_void_p()
            _lib.check(self._lib.fd_detect_graph_create(det._h, runtime.ptr(self.frames), batch, height, width, 1.0,
                                                        float(thresh), float(nms_thresh), 1 if do_round else 0,
                                                        runtime.ptr(self.dets), runtime.ptr(self.n), None,
                                                        runtime.ptr(self.status), C.byref(handle)))
            self._h = handle
        self.stale = False

    def launch(self, frames=None):
        """Replays the pass on the current stream; ``frames`` (host array or tensor of the recorded shape) is
        copied into the input buffer first.  Returns the device tensors (dets, n, status)."""
        with torch.cuda.device(self.device):
            if frames is not None:
                t = frames if isinstance(frames, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frames))
                if tuple(t.shape) != tuple(self.frames.shape) or t.dtype != torch.uint8:
                    raise ValueError("expected uint8 frames of shape %s" % (tuple(self.frames.shape),))
                self.frames.copy_(t, non_blocking=True)
            try:
                _lib.check(self._lib.fd_detect_graph_launch(self._h, runtime.stream_ptr(self.device)))
            except _lib.FaceDetError:
                self.stale = True
                raise
        return self.dets, self.n, self.status

    def detect(self, frames):
        """List of float32 [n_i,5] arrays like ``Detector.detect_batch`` (falls back to the ordinary float32
        call when the f16x3 pass had to clamp, as that does)."""
        det = self._det
        with torch.cuda.device(self.device):
            dets, n, status = self.launch(frames)
            if det._check_status(status):
                nms_thresh, thresh, do_round = self._args
                return det._in_f32(lambda: det.detect_batch(frames, nms_thresh, thresh, do_round))
            n_host, dets_host = n.cpu().numpy(), dets.cpu().numpy()
            return [dets_host[b, :n_host[b]].copy() for b in range(dets_host.shape[0])]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fd_detect_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resize(im, target_size, max_size):
    """Scale the short side to ``target_size`` unless that makes the long side exceed
    ``max_size`` (detection.py:63-78).  Returns (resized image, scale); the bilinear
    resampling runs on the GPU with OpenCV's INTER_LINEAR fixed-point arithmetic."""
    im_size_min = min(im.shape[0], im.shape[1])
    im_size_max = max(im.shape[0], im.shape[1])
    im_scale = float(target_size) / float(im_size_min)
    if np.round(im_scale * im_size_max) > max_size:
        im_scale = float(max_size) / float(im_size_max)
    return ops.resize_bilinear(im, im_scale, im_scale), im_scale


def init_detector(prefix="detection/mxnet-face-fr50", epoch=0, gpu=0):
    """detection.py:81-88.  The reference returns (arg_params, aux_params, sym, ctx) for its
    free-function callers; here the parameters live inside the returned Detector, which
    stands in for ``sym`` (min-size 24 like the reference's initialiser)."""
    det = Detector(prefix, gpu, epoch)
    config.TEST.RPN_MIN_SIZE = 24
    det.reconfigure()
    return None, None, det, gpu


def detect_face(ctx, sym, arg_params, aux_params, img_nd, nms_thresh=0.3, thresh=0.8):
    """Free-function form used by older callers in the reference
    (data_processing/ProcessVideo.py:61, CelebA.py:164): ``sym`` is the Detector from init_detector."""
    return sym.detect_face(img_nd, nms_thresh=nms_thresh, thresh=thresh)
