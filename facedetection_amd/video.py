"""Video front-end: the frame loops around the detector.

  * ``main`` -- the command-line loop of the reference's code/detection/detection.py:93-189 with
    its flags (--thresh --nms-thresh --min-size --scale --max-scale --prefix --epoch --gpu):
    frame -> RGB -> resize -> detect -> boxes mapped back to the frame -> face regions saved.
  * ``FaceFrameFilter`` / ``compress_video`` -- the "keep only the stretches with faces" logic
    of code/data_processing/ProcessVideo.py:26-78 (50-frame look-back buffer, record mode).

There is no video codec here (the reference decodes with OpenCV, which is not part of this
build): frames come from ``.npy`` files holding uint8 [N,H,W,3] (or [H,W,3]) arrays, from
binary PPM (P6) images, or from raw packed 24-bit video (``--raw-size WxH``), in BGR order as
cv2.VideoCapture delivers them.  Frames are processed in micro-batches, which changes
nothing in the per-frame results.
"""
import argparse
import collections
import os
import sys
import time

import numpy as np


# --------------------------------------------------------------------------- frame sources
def read_ppm(path):
    """Binary PPM (P6, maxval 255) -> uint8 [H,W,3] in the file's channel order."""
    with open(path, "rb") as f:
        data = f.read()
    fields, pos = [], 0
    while len(fields) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        fields.append(data[pos:end])
        pos = end
    if fields[0] != b"P6" or int(fields[3]) != 255:
        raise ValueError("%s: only binary 8-bit PPM (P6, maxval 255) is read" % path)
    w, h = int(fields[1]), int(fields[2])
    return np.frombuffer(data, dtype=np.uint8, count=h * w * 3, offset=pos + 1).reshape(h, w, 3).copy()


def write_ppm(path, img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def iter_frames(path, raw_size=None):
    """Frames uint8 [H,W,3] from a .npy file, a .ppm file, a raw packed file, or a directory of
    those (sorted by name)."""
    if os.path.isdir(path):
        for name in sorted(os.listdir(path)):
            if name.endswith((".npy", ".ppm")) or raw_size:
                yield from iter_frames(os.path.join(path, name), raw_size)
        return
    if path.endswith(".npy"):
        a = np.load(path, mmap_mode="r")
        if a.dtype != np.uint8 or a.ndim not in (3, 4) or a.shape[-1] != 3:
            raise ValueError("%s: expected uint8 [N,H,W,3] or [H,W,3]" % path)
        for fr in (a if a.ndim == 4 else a[None]):
            yield np.ascontiguousarray(fr)
    elif path.endswith(".ppm"):
        yield read_ppm(path)
    elif raw_size:
        w, h = raw_size
        a = np.memmap(path, dtype=np.uint8, mode="r")
        for i in range(a.size // (h * w * 3)):
            yield np.array(a[i * h * w * 3:(i + 1) * h * w * 3]).reshape(h, w, 3)
    else:
        raise ValueError("%s: unknown frame container (use .npy, .ppm or --raw-size)" % path)


def batches(frames, size):
    """Groups of up to ``size`` consecutive frames of equal shape."""
    group = []
    for fr in frames:
        if group and (len(group) == size or fr.shape != group[0].shape):
            yield group
            group = []
        group.append(fr)
    if group:
        yield group


# --------------------------------------------------------------------------- face-frame filter
class FaceFrameFilter:
    """Decides which frames of a stream are written out (ProcessVideo.py:45-78): a look-back
    buffer of ``buffer_size`` frames is flushed when a face appears, recording continues while
    faces keep appearing, and stops after more than ``buffer_size`` face-less frames have been
    counted since recording began (the reference never resets that count on a hit).
    ``push(frame, has_face)`` returns the frames to write now, oldest first."""

    def __init__(self, buffer_size=50):
        self.buffer_size = buffer_size
        self.buffer = collections.deque()
        self.record_mode = False
        self.missed = 0

    def _dump(self):
        out = list(self.buffer)
        self.buffer.clear()
        return out

    def push(self, frame, has_face):
        out = []
        self.buffer.append(frame)
        full = lambda: len(self.buffer) >= self.buffer_size
        if has_face:
            if not self.record_mode:
                self.record_mode = True
                out += self._dump()
            elif full():
                out += self._dump()
        elif self.record_mode:
            self.missed += 1
            if full():
                out += self._dump()
            if self.missed > self.buffer_size:
                out += self._dump()
                self.record_mode = False
                self.missed = 0
        if full():
            self.buffer.popleft()
        return out


def compress_video(frames, detector, batch=8, buffer_size=50, nms_thresh=0.3, thresh=0.8):
    """Yield the face-involved frames of a stream (ProcessVideo.py:26-78); ``frames`` is any
    iterable of BGR uint8 [H,W,3] arrays, fed to ``detect_face`` unresized like the reference."""
    filt = FaceFrameFilter(buffer_size)
    for group in batches(frames, batch):
        for fr, dets in zip(group, detector.detect_batch(np.stack(group), nms_thresh, thresh)):
            yield from filt.push(fr, dets.shape[0] > 0)


# --------------------------------------------------------------------------- command line
def parse_args(argv=None):
    p = argparse.ArgumentParser(description="face detection on a frame sequence")
    p.add_argument("--frames", type=str, required=True, help=".npy / .ppm / raw file or a directory of them")
    p.add_argument("--raw-size", type=str, default=None, help="WxH of packed 24-bit raw frames")
    p.add_argument("--rgb", action="store_true", help="frames are RGB already (default: BGR, as decoded video)")
    p.add_argument("--out", type=str, default="extracted_faces", help="directory for the face regions")
    p.add_argument("--batch", type=int, default=8, help="frames per launch")
    p.add_argument("--gpu", type=int, default=0, help="the gpu id used for predict")
    p.add_argument("--prefix", type=str, default="mxnet-face-fr50", help="the prefix of the pre-trained model")
    p.add_argument("--epoch", type=int, default=0, help="the epoch of the pre-trained model")
    p.add_argument("--thresh", type=float, default=0.8, help="the threshold of face score")
    p.add_argument("--nms-thresh", type=float, default=0.3, help="the threshold of nms")
    p.add_argument("--min-size", type=int, default=24, help="the min size of object")
    p.add_argument("--scale", type=int, default=600, help="the scale of shorter edge will be resize to")
    p.add_argument("--max-scale", type=int, default=1000, help="the maximize scale after resize")
    p.add_argument("--precision", type=str, default="f32", choices=["f32", "f16x3"],
                   help="arithmetic of the convolutions: float32 operands (the reference's) or the faster fp16 hi+lo split")
    return p.parse_args(argv)


def main(argv=None):
    from .detection import Detector
    from .symbol.config import config
    args = parse_args(argv)
    raw = tuple(int(v) for v in args.raw_size.lower().split("x")) if args.raw_size else None
    det = Detector(args.prefix, args.gpu, args.epoch, precision=args.precision)
    config.TEST.RPN_MIN_SIZE = args.min_size            # detection.py:186-188
    config.SCALES = (args.scale,)
    config.MAX_SIZE = args.max_scale
    det.reconfigure()
    os.makedirs(args.out, exist_ok=True)
    frame_id, n_faces = 0, 0
    try:
        for group in batches(iter_frames(args.frames, raw), args.batch):
            tic = time.time()
            _, boxes, _ = det.detect_scaled_batch(np.stack(group), args.scale, args.max_scale, args.nms_thresh,
                                                  args.thresh, bgr=not args.rgb)
            for color, bx in zip(group, boxes):
                for x1, y1, x2, y2 in bx:
                    # the reference names the file after the frame, so a frame's last face wins
                    # (detection.py:149-151); here every face gets its own index as well
                    region = color[max(y1, 0):max(y2, 0), max(x1, 0):max(x2, 0)]
                    if region.size:
                        write_ppm(os.path.join(args.out, "face_region_%d_%d.ppm" % (frame_id, n_faces)), region)
                    n_faces += 1
                frame_id += 1
            print("time cost is:{}s".format((time.time() - tic) / len(group)))
    finally:
        det.close()
    print("%d frames, %d face regions" % (frame_id, n_faces))
    return 0


if __name__ == "__main__":
    sys.exit(main())
