"""Consumer of the per-frame detection lists: a SORT multi-object tracker (Bewley et al.,
"Simple Online and Realtime Tracking", 2016) with the interface of the reference's
code/tracking/sort.py:

    tracker = Sort(max_age=1, min_hits=3)        # sort.py:176
    ids = tracker.update(dets)                   # sort.py:185 -> float64 [M,5], last column = id

``update`` must be called once per frame, in frame order, also with an empty ``[0,5]`` array.
This is sequential, tiny host work (a handful of 7x7 matrices per face), which the reference
also runs on the CPU; the GPU side only has to deliver the lists in order
(``parallel.gather_detections``).

Model (sort.py:75-132): state (u, v, s, r, du, dv, ds) -- box centre, area, aspect ratio and
the first three's velocities; constant-velocity transition; the measurement is (u, v, s, r).
Association (sort.py:134-171): IoU matrix between detections and predicted boxes, optimal
assignment, matches below IoU 0.3 are dissolved.  The reference's dependencies (filterpy's
Kalman filter, scikit-learn's removed ``linear_assignment_``) are replaced by the closed-form
filter equations below and ``scipy.optimize.linear_sum_assignment``.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment


def iou(bb_test, bb_gt):
    """IoU of two boxes (x1,y1,x2,y2), extents without the +1 pixel convention (sort.py:33-46)."""
    w = max(0.0, min(bb_test[2], bb_gt[2]) - max(bb_test[0], bb_gt[0]))
    h = max(0.0, min(bb_test[3], bb_gt[3]) - max(bb_test[1], bb_gt[1]))
    inter = w * h
    return inter / ((bb_test[2] - bb_test[0]) * (bb_test[3] - bb_test[1]) +
                    (bb_gt[2] - bb_gt[0]) * (bb_gt[3] - bb_gt[1]) - inter)


def iou_matrix(dets, trks):
    """float32 [len(dets), len(trks)] of pairwise IoUs (the double loop at sort.py:142-146)."""
    d = np.asarray(dets, dtype=np.float64)[:, None, :4]
    t = np.asarray(trks, dtype=np.float64)[None, :, :4]
    w = np.maximum(0.0, np.minimum(d[..., 2], t[..., 2]) - np.maximum(d[..., 0], t[..., 0]))
    h = np.maximum(0.0, np.minimum(d[..., 3], t[..., 3]) - np.maximum(d[..., 1], t[..., 1]))
    inter = w * h
    union = (d[..., 2] - d[..., 0]) * (d[..., 3] - d[..., 1]) + (t[..., 2] - t[..., 0]) * (t[..., 3] - t[..., 1]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / union).astype(np.float32)


def convert_bbox_to_z(bbox):
    """(x1,y1,x2,y2) -> column (cx, cy, area, w/h) (sort.py:48-60)."""
    w = bbox[2] - bbox[0]
    h = bbox[3] - bbox[1]
    return np.array([bbox[0] + w / 2.0, bbox[1] + h / 2.0, w * h, w / float(h)], dtype=np.float64).reshape(4, 1)


def convert_x_to_bbox(x, score=None):
    """State column -> row (x1,y1,x2,y2[,score]) (sort.py:62-72)."""
    w = np.sqrt(x[2] * x[3])
    h = x[2] / w
    row = [x[0] - w / 2.0, x[1] - h / 2.0, x[0] + w / 2.0, x[1] + h / 2.0]
    if score is not None:
        row.append(np.atleast_1d(score))
    return np.concatenate([np.atleast_1d(v).astype(np.float64).ravel() for v in row]).reshape(1, -1)


_F = np.eye(7)
_F[0, 4] = _F[1, 5] = _F[2, 6] = 1.0
_H = np.eye(4, 7)


class KalmanBoxTracker:
    """One tracked face (sort.py:75-132).  Ids are handed out from a process-wide counter
    like the reference's class attribute, so they stay unique across trackers."""
    count = 0

    def __init__(self, bbox):
        # noise levels of sort.py:89-93 on identity defaults
        self.R = np.diag([1.0, 1.0, 10.0, 10.0])
        self.P = np.diag([10.0, 10.0, 10.0, 10.0, 1e4, 1e4, 1e4])
        self.Q = np.diag([1.0, 1.0, 1.0, 1.0, 0.01, 0.01, 1e-4])
        self.x = np.zeros((7, 1))
        self.x[:4] = convert_bbox_to_z(bbox)
        self.time_since_update = 0
        self.id = KalmanBoxTracker.count
        KalmanBoxTracker.count += 1
        self.history = []
        self.hits = 0
        self.hit_streak = 0
        self.age = 0

    def update(self, bbox):
        """Measurement update with an observed box (sort.py:104-112)."""
        self.time_since_update = 0
        self.history = []
        self.hits += 1
        self.hit_streak += 1
        z = convert_bbox_to_z(bbox)
        y = z - _H @ self.x
        PHt = self.P @ _H.T
        S = _H @ PHt + self.R
        K = PHt @ np.linalg.inv(S)
        self.x = self.x + K @ y
        # Joseph form keeps P symmetric positive definite
        IKH = np.eye(7) - K @ _H
        self.P = IKH @ self.P @ IKH.T + K @ self.R @ K.T

    def predict(self):
        """Time update; returns the predicted box [1,4] (sort.py:114-126)."""
        if (self.x[6] + self.x[2]) <= 0:
            self.x[6] *= 0.0
        self.x = _F @ self.x
        self.P = _F @ self.P @ _F.T + self.Q
        self.age += 1
        if self.time_since_update > 0:
            self.hit_streak = 0
        self.time_since_update += 1
        self.history.append(convert_x_to_bbox(self.x))
        return self.history[-1]

    def get_state(self):
        return convert_x_to_bbox(self.x)


def associate_detections_to_trackers(detections, trackers, iou_threshold=0.3):
    """(matches [m,2], unmatched detection indices, unmatched tracker indices) (sort.py:134-171)."""
    if len(trackers) == 0:
        return np.empty((0, 2), dtype=int), np.arange(len(detections)), np.empty((0,), dtype=int)
    if len(detections) == 0:
        return np.empty((0, 2), dtype=int), np.empty((0,), dtype=int), np.arange(len(trackers))
    m = iou_matrix(detections, trackers)
    rows, cols = linear_sum_assignment(-np.nan_to_num(m, nan=0.0))
    unmatched_d = [d for d in range(len(detections)) if d not in rows]
    unmatched_t = [t for t in range(len(trackers)) if t not in cols]
    matches = []
    for d, t in zip(rows, cols):
        if m[d, t] < iou_threshold:
            unmatched_d.append(d)
            unmatched_t.append(t)
        else:
            matches.append((d, t))
    matches = np.array(matches, dtype=int).reshape(-1, 2)
    return matches, np.array(unmatched_d, dtype=int), np.array(unmatched_t, dtype=int)


class Sort:
    def __init__(self, max_age=1, min_hits=3):
        self.max_age = max_age
        self.min_hits = min_hits
        self.trackers = []
        self.frame_count = 0

    def update(self, dets):
        """dets [N,5] (x1,y1,x2,y2,score), possibly [0,5] -> float64 [M,5] (x1,y1,x2,y2,id);
        ids start at 1; newest track first (sort.py:185-230)."""
        dets = np.asarray(dets, dtype=np.float64).reshape(-1, 5)
        self.frame_count += 1
        predicted, alive = [], []
        for trk in self.trackers:
            pos = trk.predict()[0]
            if np.any(np.isnan(pos)):
                continue                      # a degenerate state drops the track
            predicted.append(pos)
            alive.append(trk)
        self.trackers = alive
        trks = np.array(predicted, dtype=np.float64).reshape(-1, 4)
        matched, unmatched_dets, _ = associate_detections_to_trackers(dets, trks)
        for d, t in matched:
            self.trackers[t].update(dets[d])
        for i in unmatched_dets:
            self.trackers.append(KalmanBoxTracker(dets[i]))
        ret, keep = [], []
        for trk in reversed(self.trackers):
            if trk.time_since_update < 1 and (trk.hit_streak >= self.min_hits or self.frame_count <= self.min_hits):
                ret.append(np.concatenate((trk.get_state()[0], [trk.id + 1])).reshape(1, -1))
            if trk.time_since_update <= self.max_age:
                keep.append(trk)
        self.trackers = keep[::-1]
        if ret:
            return np.concatenate(ret)
        return np.empty((0, 5))
