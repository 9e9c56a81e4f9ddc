"""The sharded path end to end on real detections: three (or, on 4K frames, two) processes share the one GPU of
the test box, each runs its own detector on the frames i with i mod world == rank (the reference's process-per-GPU scheme,
code/data_processing/ProcessVideo.py:92-114, with frames instead of videos as the unit), the
variable-length detection lists are all-gathered (gloo here: RCCL does not accept several ranks on one
device; the collective calls are the same) and rank 0's gathered lists must equal, frame by frame and
bit for bit, what one process computes alone -- a frame's detections do not depend on how frames were
batched or sharded.  The gathered lists then drive a SORT tracker in global frame order."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {"small": (8, 3, 360, 640),              # shards of 3, 3 and 2 frames
         "4k": (5, 2, 2160, 3840)}               # BASELINE configuration 4's frame size: shards of 3 and 2 frames


def _frame(i, h, w):
    from facedetection_amd.synthetic import synthetic_image
    if (h, w) == (2160, 3840):
        return np.tile(synthetic_image(1080, 1920, seed=i), (2, 2, 1))
    return synthetic_image(h, w, seed=i)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, case, precision):
    import torch
    import torch.distributed as dist
    from facedetection_amd.detection import Detector
    from facedetection_amd.parallel import gather_detections, shard_indices
    N_FRAMES, _, H, W = CASES[case]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_indices(N_FRAMES, rank, world)
        det = Detector("synthetic:0", 0, precision=precision)
        try:
            frames = torch.from_numpy(np.stack([_frame(i, H, W) for i in mine])).cuda()
            dets, counts, status = det.detect_batch_device(frames, thresh=0.5)
            torch.cuda.synchronize()
            assert int(status[0].item()) == 0
        finally:
            det.close()
        out = gather_detections(dets.cpu(), counts.cpu(), rank, world)
        if rank == 0:
            np.save(os.path.join(out_dir, "len.npy"), np.array([len(o) for o in out], dtype=np.int64))
            np.save(os.path.join(out_dir, "rows.npy"),
                    np.concatenate([o.reshape(-1, 5) for o in out] + [np.zeros((0, 5), np.float32)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("case", ["small", "4k"])
def test_processes_sharing_one_gpu_gather_what_one_process_detects(tmp_path, case, precision):
    import torch.multiprocessing as mp
    from facedetection_amd.detection import Detector
    from facedetection_amd.parallel import feed_tracker
    from facedetection_amd.tracking.sort import Sort
    N_FRAMES, WORLD, H, W = CASES[case]
    port = _free_port()
    ctx = mp.get_context("spawn")                 # fresh processes: nothing inherited from this one's GPU state
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, str(tmp_path), case, precision)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    lens = np.load(tmp_path / "len.npy")
    rows = np.load(tmp_path / "rows.npy")
    gathered = np.split(rows, np.cumsum(lens)[:-1])
    det = Detector("synthetic:0", 0, precision=precision)
    try:
        alone = [det.detect_face(_frame(i, H, W), thresh=0.5) for i in range(N_FRAMES)]
    finally:
        det.close()
    assert len(gathered) == N_FRAMES and sum(len(a) for a in alone) > 0
    for i in range(N_FRAMES):
        np.testing.assert_array_equal(gathered[i].reshape(-1, 5), alone[i])
    # the hand-off to the order-dependent consumer gives the same tracks either way
    # (ids come from a process-wide counter, as in the reference's tracker: restart it for each run)
    from facedetection_amd.tracking.sort import KalmanBoxTracker
    saved = KalmanBoxTracker.count
    try:
        KalmanBoxTracker.count = 0
        t_a = feed_tracker(Sort(), [g.reshape(-1, 5) for g in gathered])
        KalmanBoxTracker.count = 0
        t_b = feed_tracker(Sort(), alone)
    finally:
        KalmanBoxTracker.count = saved
    assert sum(len(a) for a in t_a) > 0
    for a, b in zip(t_a, t_b):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def _nccl_worker(port, out_dir):
    """One rank on the RCCL backend: the detection tensors stay on the device through gather_detections
    (count exchange and padded gather as device collectives)."""
    import torch
    import torch.distributed as dist
    from facedetection_amd.detection import Detector
    from facedetection_amd.parallel import gather_detections
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        det = Detector("synthetic:0", 0)
        try:
            frames = torch.from_numpy(np.stack([_frame(i, 360, 640) for i in range(4)])).cuda()
            dets, counts, status = det.detect_batch_device(frames, thresh=0.5)
            assert dets.is_cuda and counts.is_cuda
            out = gather_detections(dets, counts, 0, 1, force_collective=True)
            plain = gather_detections(dets, counts, 0, 1)
        finally:
            det.close()
        assert len(out) == len(plain) == 4 and sum(len(o) for o in out) > 0
        for a, b in zip(out, plain):
            np.testing.assert_array_equal(a, b)
        np.save(os.path.join(out_dir, "ok.npy"), np.array([len(o) for o in out], dtype=np.int64))
    finally:
        dist.destroy_process_group()


def test_device_tensors_through_the_rccl_gather(tmp_path):
    """The device-tensor path of parallel.gather_detections on hardware (backend "nccl" = RCCL), one rank: the
    collectives run on the GPU and give what the collective-free path gives."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), str(tmp_path)))
    p.start()
    p.join(300)
    assert p.exitcode == 0, p.exitcode
    assert np.load(tmp_path / "ok.npy").sum() > 0
