"""The N>1 hand-off on CPU: two gloo ranks shard frames by index and all-gather their
variable-length detection lists back into global frame order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from facedetection_amd.parallel import gather_detections, shard_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_detections(global_frame, R=10):
    """Deterministic stand-in for a detector: frame i has (i*3) % (R+1) boxes."""
    rng = np.random.default_rng(global_frame)
    n = (global_frame * 3) % (R + 1)
    dets = np.zeros((R, 5), np.float32)
    dets[:n] = rng.uniform(0, 500, (n, 5)).astype(np.float32)
    return dets, n


def _worker(rank, world, port, n_frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(n_frames, rank, world)
    d = np.stack([_fake_detections(i)[0] for i in mine]) if mine else np.zeros((0, 10, 5), np.float32)
    c = np.array([_fake_detections(i)[1] for i in mine], np.int32)
    out = gather_detections(torch.from_numpy(d), torch.from_numpy(c), rank, world)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank),
            np.concatenate([o.reshape(-1) for o in out] + [np.zeros(0, np.float32)]))
    np.save(os.path.join(out_dir, "len%d.npy" % rank), np.array([len(o) for o in out], dtype=np.int64))
    dist.destroy_process_group()


def test_shard_indices_cover_every_frame_once():
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in shard_indices(37, r, world))
        assert seen == list(range(37))


def test_single_rank_gather_is_a_trim():
    d = np.stack([_fake_detections(i)[0] for i in range(5)])
    c = np.array([_fake_detections(i)[1] for i in range(5)], np.int32)
    out = gather_detections(torch.from_numpy(d), torch.from_numpy(c))
    assert [len(o) for o in out] == list(c)
    np.testing.assert_array_equal(out[3], d[3, :c[3]])


@pytest.mark.parametrize("world,n_frames", [(2, 12), (2, 13), (3, 14), (3, 2), (2, 0)])
def test_gloo_gather_restores_global_order(tmp_path, world, n_frames):
    """Even shards, shards of unequal length (13 frames on 2 ranks, 14 on 3), fewer frames than
    ranks, and no frames at all."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_frames, str(tmp_path)), nprocs=world, join=True)
    expect = [_fake_detections(i) for i in range(n_frames)]
    flat = np.concatenate([d[:n].reshape(-1) for d, n in expect] + [np.zeros(0, np.float32)])
    lens = np.array([n for _, n in expect], dtype=np.int64)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / ("rank%d.npy" % r)), flat)
        np.testing.assert_array_equal(np.load(tmp_path / ("len%d.npy" % r)), lens)
