"""bench.py prints exactly one JSON line with the fields the benchmark contract names: a float32
headline with roofline and cpu_baseline, the sub-records of the other configurations (single-frame latency,
4K frames, dense crowd, front-end resize) with the HBM fractions of the memory-bound stages, and on request
the f16x3 sub-record with its measured error."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_roofline(r):
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic FLOP per launch / average launch duration, both in the record
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) <= 2e-3 * r["achieved"]
    # traffic is either a measurement taken on this very kernel set or null with the reason
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["traffic_note"]


def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "16", "--steps", "1", "--warmup", "1",
                          "--check-frames", "1", "--direct-record", "--split-record"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "f16x3", "accuracy_vs_float64",
                "n_ranks_seen"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1 and d["steps"] == 1 and d["warmup"] == 1
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["dtype"] == "f32"                                   # the headline is the reference's arithmetic
    assert d["value"] > 0 and abs(d["value"] - 16 / (d["ms_per_step"] / 1e3)) <= 1e-3 * d["value"]
    _check_roofline(d["roofline"])
    assert d["roofline"]["peak"] == 157.3 and d["roofline"]["kernel"].startswith("conv_f32")
    # definition FLOPs (every layer as a direct convolution) and the FLOPs the kernels executed
    assert 0 < d["roofline"]["executed_gflop_per_frame"] <= 354.5 * 1.02
    assert 0 < d["roofline"]["pipeline_executed_frac"] < 1
    assert d["roofline"]["pipeline_executed_tflops"] <= d["roofline"]["pipeline_direct_equivalent_tflops"] * 1.02
    # the memory-bound stages carry their fraction of the HBM peak
    for name in ("proposal_decode", "roipool", "nms_mask+nms_reduce"):
        r = d["hbm_bound_stages"][name]
        assert r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3, name
    # the other configurations of the benchmark definition, same precision, same process
    c = d["configs"]
    assert 0 < c["latency_1080p"]["ms_back_to_back"] <= c["latency_1080p"]["ms"] * 1.05 and c["latency_1080p"]["ms"] < 20
    assert c["frame_4k"]["value"] > 0 and c["frame_4k"]["ms_single_frame"] > c["latency_1080p"]["ms_back_to_back"]
    dc = c["dense_crowd"]
    assert dc["rois_into_stage4"] == 6000 and dc["candidates_into_rpn_nms"] >= 90000 and dc["ms"] > 0
    assert "nms_mask+nms_reduce" in dc["hbm_bound_stages"]
    rz = c["frontend_resize"]
    assert rz["bound"] == "hbm" and rz["algorithmic_bytes"] == 16 * 3 * (1080 * 1920 + 562 * 1000)
    assert abs(rz["frac"] - rz["achieved"] / rz["peak"]) < 1e-3 and 0 < rz["frac"] < 1
    # traffic of the whole float32 pass against the HBM peak: a measurement on this kernel set, or null with the reason
    h = d["hbm_whole_pass"]
    assert h["note"] and (h["bytes_per_frame"] is None or
                          (h["peak"] == 8000.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-3 and
                           abs(h["achieved"] - h["bytes_per_frame"] * d["value"] / 1e9) <= 1e-3 * h["achieved"] + 0.1))
    a = d["accuracy_vs_float64"]
    assert a["max_abs_score_err_vs_f64"] <= 1e-4 and a["max_box_err_px"] <= 0.5 and a["clamped_values"] == 0
    s = d["f16x3"]
    for key in ("value", "ms_per_step", "roofline", "max_abs_score_err_vs_f64", "max_box_err_px", "clamped_values"):
        assert key in s, key
    assert s["value"] > d["value"]
    assert s["max_abs_score_err_vs_f64"] <= 1e-4 and s["max_box_err_px"] <= 0.5
    assert s["clamped_values"] == 0 and s["clamped_values_last_timed_step"] == 0
    _check_roofline(s["roofline"])
    # the opt-in three-way bf16 split: float32 tensors and range, its own kernels and matrix peak (2500 / 6)
    b = d["bf16x3"]
    assert b["value"] > 0 and b["clamped_values"] == 0 and "bf16" in b["dtype"]
    assert b["max_abs_score_err_vs_f64"] <= 1e-4 and b["max_box_err_px"] <= 0.5
    _check_roofline(b["roofline"])
    assert b["roofline"]["kernel"].startswith("conv_b3") and abs(b["roofline"]["peak"] - 2500.0 / 6) < 0.1
    # the same float32 pass without the Winograd form of the wide 3x3 layers, reported next to the headline
    dc = d["f32_direct_convolutions"]
    assert dc["dtype"] == "f32" and 0 < dc["value"] <= d["value"] * 1.05
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1


def test_asking_for_more_gpus_than_visible_fails_loudly():
    import torch
    want = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode == 2 and out.stdout.strip() == "" and "GPU(s) visible" in out.stderr


def test_one_rank_under_the_distributed_launcher_over_rccl():
    """The launch path of the multi-GPU runs with a single rank: torch.distributed.run starts bench.py, the process
    group comes up on the "nccl" (RCCL) backend, barriers bracket the timed steps, the detection lists go through the
    all-gathers of parallel.gather_detections on device tensors, and stdout still carries exactly one JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--frames", "16", "--steps", "1", "--warmup", "1", "--check-frames", "0",
                          "--no-config-records", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                         env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1 and d["value"] > 0 and d["dtype"] == "f32"
    assert d["config"]["detections_in_last_step"] >= 0
    _check_roofline(d["roofline"])
