"""bench.py's own launcher (`--gpus N` without torch.distributed.run), rehearsed on CPU over gloo:
same rank start-up, barriers, max-over-ranks timing and detection gather as the GPU run, with a
stand-in for the detector.  And the ways it must refuse to run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return e


def test_two_ranks_started_by_bench_itself():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse", "--frames", "7", "--steps", "2",
                          "--warmup", "1"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["rehearsal"] is True and d["value"] is None          # nothing was measured, and the line says so
    assert d["frames_gathered"] == 14 and d["boxes_gathered"] == d["boxes_expected"] > 0


def test_three_ranks_with_a_ragged_shard():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--rehearse", "--frames", "5", "--steps", "1",
                          "--warmup", "0"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_ranks_seen"] == 3 and d["frames_gathered"] == 15 and d["boxes_gathered"] == d["boxes_expected"]


def test_more_gpus_than_the_host_has_fails_loudly():
    """Without --rehearse the launcher counts the visible GPUs first: asking for more is an error
    (exit code 2, a message, no JSON line) -- never a silent single-GPU run."""
    import torch
    want = torch.cuda.device_count() + 1
    if want < 2:
        want = 2
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(want), "--steps", "1", "--warmup", "0"], cwd=ROOT,
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 2
    assert out.stdout.strip() == "" and "GPU(s) visible" in out.stderr


def test_gpus_flag_must_match_the_launcher():
    e = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse"], cwd=ROOT, env=e, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 2 and "must agree" in out.stderr and out.stdout.strip() == ""
