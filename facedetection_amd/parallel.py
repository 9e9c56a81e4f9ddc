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
    [n_i,5] arrays, ready for an order-dependent consumer such as a SORT tracker.

    Ranks may hold different numbers of frames (a frame count that is not a multiple of the
    world size gives the first ranks one frame more): the per-rank frame counts are exchanged
    first and every rank pads its tensors to the longest shard before the two gathers, so all
    collectives see equal sizes on every rank."""
    F, R = int(dets.shape[0]), int(dets.shape[1])
    if world == 1 and not force_collective:
        d, c = dets.cpu().numpy(), counts.cpu().numpy()
        return [d[i, :c[i]].copy() for i in range(F)]
    import torch.distributed as dist
    dev = dets.device
    # 1. frames per rank
    mine = torch.tensor([F], dtype=torch.int64, device=dev)
    all_f = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(all_f, mine)
    fs = [int(t.item()) for t in all_f]
    n_frames, f_max = sum(fs), max(fs)
    for r in range(world):
        if fs[r] != len(range(r, n_frames, world)):
            raise ValueError("rank %d holds %d frames; frame i -> rank i mod %d of %d frames gives %d"
                             % (r, fs[r], world, n_frames, len(range(r, n_frames, world))))
    if f_max == 0:
        return []
    # 2. detections per frame, padded to the longest shard
    c_pad = torch.zeros((f_max,), dtype=torch.int32, device=dev)
    c_pad[:F] = counts.to(torch.int32)
    all_counts = [torch.empty_like(c_pad) for _ in range(world)]
    dist.all_gather(all_counts, c_pad)
    # 3. the rows, trimmed to the longest list in the job
    max_n = min(R, max(1, int(torch.stack(all_counts).max().item())))
    payload = torch.zeros((f_max, max_n, 5), dtype=torch.float32, device=dev)
    payload[:F] = dets[:, :max_n, :]
    all_dets = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(all_dets, payload)
    cs = [c.cpu().numpy() for c in all_counts]
    ds = [d.cpu().numpy() for d in all_dets]
    out = []
    for i in range(n_frames):
        r, j = i % world, i // world
        out.append(ds[r][j, :cs[r][j]].copy())
    return out


def feed_tracker(tracker, dets_per_frame):
    """Hand the gathered lists to an order-dependent consumer: ``tracker.update`` is called once
    per frame in global frame order, empty frames included (the contract of
    code/tracking/sort.py:185-193).  Returns the tracker's output per frame."""
    return [tracker.update(d) for d in dets_per_frame]
