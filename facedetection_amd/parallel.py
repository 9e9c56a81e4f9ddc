"""Multi-GPU hand-off: one process per GPU, frames sharded by index, no collective on the
data path.  The only exchange is the all-gather of the variable-length detection lists
(a count exchange, then one gather padded to the longest list) -- RCCL over xGMI when the
process group uses the "nccl" backend, gloo on CPU for tests.  Messages are tiny (<= 200 B
per frame at the default 10 rois), so this is latency-bound and done once per batch.

The reference's own scheme is coarser: independent processes pulling whole videos from a
flock-guarded list (code/data_processing/ProcessVideo.py:92-114)."""
import numpy as np
import torch


def shard_indices(n_frames, rank, world):
    """Global indices of the frames rank ``rank`` processes: frame i -> rank i mod world."""
    return list(range(rank, n_frames, world))


def gather_detections(dets, counts, rank=0, world=1, force_collective=False):
    """dets [F,R,5] float32 and counts [F] int32 of this rank's frames (frame j of rank r is
    global frame j*world + r) -> on every rank, a list over GLOBAL frame order of float32
    [n_i,5] arrays, ready for an order-dependent consumer such as a SORT tracker."""
    F, R = int(dets.shape[0]), int(dets.shape[1])
    if world == 1 and not force_collective:
        d, c = dets.cpu().numpy(), counts.cpu().numpy()
        return [d[i, :c[i]].copy() for i in range(F)]
    import torch.distributed as dist
    # counts first (fixed size), then the rows trimmed to the longest list in the job
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts.contiguous())
    max_n = max(1, int(torch.stack(all_counts).max().item()))
    payload = dets[:, :max_n, :].contiguous()
    all_dets = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(all_dets, payload)
    cs = [c.cpu().numpy() for c in all_counts]
    ds = [d.cpu().numpy() for d in all_dets]
    out = []
    for j in range(F):
        for r in range(world):
            out.append(ds[r][j, :cs[r][j]].copy())
    return out


def feed_tracker(tracker, dets_per_frame):
    """Hand the gathered lists to an order-dependent consumer: ``tracker.update`` is called once
    per frame in global frame order, empty frames included (the contract of
    code/tracking/sort.py:185-193).  Returns the tracker's output per frame."""
    return [tracker.update(d) for d in dets_per_frame]
