#!/usr/bin/env python3
"""Headline benchmark: 1080p frames/sec of the whole detection hot path on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--batch B]

One "step" = one pass of Detector.detect_face's pipeline (input staging, ResNet-50 backbone,
RPN, proposal, ROI pooling, stage-4 head, decode/threshold/NMS/round) over F synthetic
1920x1080 frames that are already resident in HBM, issued in micro-batches of B frames.
Frames are sharded by index across ranks (one process per GPU, no data-path collective:
"weak" scaling -- each rank owns F frames); the detection lists are all-gathered after the
timed region only to exercise the hand-off.  Rank 0 prints ONE JSON line.

The line's headline (`value`, `dtype`, `roofline`) is measured with float32 operands on the fp32
matrix cores -- the reference's arithmetic.  After the timed region the same process measures, as
sub-records under `configs`, the other configurations of the benchmark definition at the same
precision (single 1080p frame latency, 4K frames, the dense-crowd stress setting) and the
front-end resize, each HBM-bound stage with its fraction of the HBM peak, and as `bf16x3` the opt-in
three-way bf16 split (float32 tensors and range, products on the bf16 matrix cores; own kernels, own
roofline entry); `--split-record` adds the opt-in f16x3 arithmetic (not the reference's: 22 significant
bits, fp16 range) as `f16x3`.

Launching: `python bench.py --gpus N` with N > 1 and no RANK in the environment starts N ranks
itself (fresh processes, one per GPU, before this process touches a GPU); under
`python -m torch.distributed.run` the ranks it is given are used.  `--rehearse` runs the same
launcher, barriers, timing and gather on CPU over gloo with a stand-in for the detector (no
measurement: the line says so).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 1080, 1920               # the headline's frame size (--height / --width)


def gmac_per_frame(height, width, rois=10):
    """Multiply-accumulates of one frame by the graph's definition (every layer a direct convolution;
    code/detection/symbol/faster_rcnn.py:107-162): stem 7x7/2, max-pool, bottleneck stages of (3, 4, 6)
    units to 1024 channels at stride 16, the RPN's 3x3 + 1x1 heads, and per roi the 3 units of stage 4 on a
    7x7 map plus the two FC layers.  1080p with 10 rois: 177.25 GMAC."""
    def out(n, k, s, p):
        return (n + 2 * p - k) // s + 1
    h, w = out(height, 7, 2, 3), out(width, 7, 2, 3)
    macs = h * w * 147 * 64
    h, w = out(h, 3, 2, 1), out(w, 3, 2, 1)
    cin = 64
    for s, (units, c) in enumerate(((3, 256), (4, 512), (6, 1024))):
        mid = c // 4
        for u in range(units):
            st = 2 if (u == 0 and s > 0) else 1
            ho, wo = out(h, 3, st, 1), out(w, 3, st, 1)
            macs += h * w * cin * mid + ho * wo * 9 * mid * mid + ho * wo * mid * c
            if u == 0:
                macs += ho * wo * cin * c
            h, w, cin = ho, wo, c
    macs += h * w * (9 * 1024 * 512 + 512 * 72)
    head = 49 * 1024 * 512 + 16 * 9 * 512 * 512 + 16 * 512 * 2048 + 16 * 1024 * 2048
    head += 2 * (16 * 2048 * 512 + 16 * 9 * 512 * 512 + 16 * 512 * 2048) + 2048 * 10
    return (macs + rois * head) / 1e9


GFLOP_PER_FRAME = 354.5          # 1080p, post_nms_top_n = 10: 2 x gmac_per_frame(1080, 1920) = 2 x 177.25 GMAC
PEAK_FP32_MATRIX_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MATRIX_TFLOPS = 2500.0  # v_mfma_f32_16x16x32_f16, dense
PEAK_HBM_GBS = 8000.0            # HBM3E peak
B3_DTYPE = ("f32 tensors; every product from exact three-way bf16 splits of both operands (6 bf16 MFMAs per product, "
            "f32 accumulate, f32 exponent range)")
PROP = dict(pre_nms_top_n=1000, post_nms_top_n=10, nms_thresh=0.7, min_size=50)
PMC_BATCH = 16                   # default micro-batch; the committed PMC passes are taken at it (16 frames' stage-1
                                 # tensors are the largest the 2^31-byte buffer-addressed kernels reach)


def make_frames(n, device, seed):
    """n distinct uint8 [1080,1920,3] frames on the device from 4 CPU-generated base frames
    (shifted and re-noised on the GPU; data generation only, not part of the measured path)."""
    from facedetection_amd.synthetic import synthetic_image
    base = np.stack([synthetic_image(H, W, seed=seed * 100 + i) for i in range(4)])
    base = torch.from_numpy(base).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device=device)
    for i in range(n):
        f = torch.roll(base[i % 4], shifts=(7 * i % H, 13 * i % W), dims=(0, 1)).to(torch.int16)
        f = f + torch.randint(-6, 7, f.shape, generator=g, device=device, dtype=torch.int16)
        out[i] = f.clamp_(0, 255).to(torch.uint8)
    return out


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (FETCH_SIZE and WRITE_SIZE, separate runs of this benchmark at the same micro-batch;
    profiles/*pmc_traffic.json) -- but only if that file was taken on the kernel set of the
    library that is running (its csrc_sha16 equals fd_build_id()).  Returns (bytes or None, note).
    Counters cannot be read from inside this process."""
    import glob
    import re
    from facedetection_amd import _lib
    build = _lib.lib().fd_build_id().decode("ascii")
    m = re.match(r"conv_(split|f32|igemm)_(\d+)x(\d+)(_pre|_o\d)?(_bn)?$", kernel)
    # (the passes over the front-end benchmark, *_frontend_pmc_traffic.json, do not cover this benchmark's kernels)
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json"))
                   if "frontend" not in os.path.basename(f))
    if not m or not files:
        return None, "no PMC pass covers this kernel"
    with open(files[-1]) as f:
        doc = json.load(f)
    name = os.path.basename(files[-1])
    if doc.get("csrc_sha16") != build:
        return None, ("%s was taken on kernel set %s, the library running is %s: refresh the PMC passes"
                      % (name, doc.get("csrc_sha16", "(unrecorded)"), build))
    family, bm, bn, pre = m.group(1), m.group(2), m.group(3), m.group(4) == "_pre"
    occ = m.group(4)[2:] if (m.group(4) or "").startswith("_o") else "0"    # workgroups per CU asked for (0: default)
    if family == "igemm":
        want = re.compile(r"conv_igemm_kernel<%s, %s, false>" % (bm, bn))
    elif m.group(5):   # direct-to-LDS kernel with BatchNorm on the fragments: <BM, BN, stages, workgroups per CU, channels kept>
        want = re.compile(r"conv_split_glds_pre_kernel<%s, %s, \d+, %s, \d+>" % (bm, bn, occ))
    elif pre:       # BatchNorm-on-load kernel: <BM, BN, mode 0, float32 operands?>
        want = re.compile(r"conv_split_reg_kernel<%s, %s, 0, %s(, false)?>" % (bm, bn, "true" if family == "f32" else "false"))
    else:           # direct-to-LDS kernel: <BM, BN, stages, waves, float32 operands?, workgroups per CU, bf16x3 products?>
        want = re.compile(r"conv_split_glds_kernel<%s, %s, \d+, \d+, %s, %s(, false)?>" % (bm, bn, "true" if family == "f32" else "false", occ))
    for kname, row in doc["kernels"].items():
        if want.search(kname):
            return row["hbm_bytes_per_launch"], "HBM bytes per launch from %s (kernel set %s)" % (name, build)
    return None, "%s has no row for this kernel" % name


def hbm_whole_pass(fps):
    """HBM traffic of the whole float32 pass per frame (every kernel, PMC counters: profiles/*pmc_traffic_f32.json,
    two passes over this benchmark with --no-split-record) times the measured frame rate, against the 8 TB/s peak --
    the pass as a whole is bound by the fp32 matrix cores, and this says how far from the memory roof it runs.
    Only quoted when the file was taken on the kernel set of the library that is running."""
    import glob
    from facedetection_amd import _lib
    build = _lib.lib().fd_build_id().decode("ascii")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic_f32.json")))
    if not files:
        return {"bytes_per_frame": None, "note": "no whole-pass PMC file under profiles/"}
    with open(files[-1]) as f:
        doc = json.load(f)
    name = os.path.basename(files[-1])
    whole = doc.get("whole_run")
    if doc.get("csrc_sha16") != build or not whole:
        return {"bytes_per_frame": None,
                "note": "%s was taken on kernel set %s, the library running is %s" % (name, doc.get("csrc_sha16"), build)}
    gbs = whole["hbm_bytes_per_frame"] * fps / 1e9
    return {"bytes_per_frame": whole["hbm_bytes_per_frame"], "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(gbs / PEAK_HBM_GBS, 4),
            "note": "PMC bytes of every kernel of the float32 pass per frame (%s, %d frames at micro-batch %d) x frames/s"
                    % (name, whole["frames"], PMC_BATCH)}


def cpu_baseline(budget_s=15.0):
    """The CPU oracle (torch-CPU float32 port of the same pipeline) on 1080p frames, on this
    host's cores: at least 2 frames, more until the time budget is used."""
    from facedetection_amd.synthetic import synthetic_image, synthetic_params
    from oracle import network as onet
    params = synthetic_params(0)
    threads = torch.get_num_threads()
    img = synthetic_image(H, W, seed=1)
    onet.detect_face(synthetic_image(270, 480, seed=1), params, PROP)     # warm up the thread pool
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < budget_s and n < 64):
        onet.detect_face(img, params, PROP)
        n += 1
    dt = time.time() - t0
    return dict(value=round(n / dt, 4), unit="frames/s", cores=threads, kind="port",
                sample="%d synthetic 1080p frames through oracle.network.detect_face (torch-CPU float32, %d threads), %.1f s"
                       % (n, threads, dt))


def accuracy_vs_float64(pipe, frames, n_check, precisions=("f32",)):
    """Untimed: the graph outputs of the first `n_check` benchmark frames in the given precisions against
    oracle.network in float64 on the host.  Score error = max |cls_prob - oracle|; box error = max
    coordinate difference (pixels) of the decoded, clipped class-1 boxes (detection.py:47-49)."""
    from facedetection_amd.synthetic import synthetic_params
    from oracle import boxes as ob
    from oracle import network as onet
    params = synthetic_params(0)
    det = pipe.detectors[0]
    imgs = frames[:n_check].cpu().numpy()
    refs = []
    with torch.no_grad():
        for img in imgs:
            rois, cls, deltas, _ = onet.forward(img, params, PROP, torch.float64)
            refs.append((cls, ob.clip_boxes(ob.bbox_pred(rois[:, 1:], deltas), (H, W))[:, 4:8]))
    out = {}
    saved = det.on_clamp
    det.on_clamp = "ignore"                # measure the split arithmetic itself, not its fallback
    try:
        for prec in precisions:
            det.set_precision(prec)
            s_err = b_err = 0.0
            clamped = 0
            for img, (cls_ref, box_ref) in zip(imgs, refs):
                rois, cls, deltas = det.forward(img)
                clamped += det.clamped_values()
                box = ob.clip_boxes(ob.bbox_pred(rois[:, 1:], deltas), (H, W))[:, 4:8]
                s_err = max(s_err, float(np.abs(cls - cls_ref).max()))
                b_err = max(b_err, float(np.abs(box - box_ref).max()))
            out[prec] = dict(max_abs_score_err_vs_f64=float("%.3g" % s_err), max_box_err_px=float("%.3g" % b_err),
                             clamped_values=clamped, frames_checked=int(n_check))
    finally:
        det.on_clamp = saved
    return out


def _kernel_table(rows, calls=1):
    """Per-kernel rows of a bracketed run: kernels with FLOPs carry TFLOP/s, every kernel its algorithmic
    GB/s and that rate as a fraction of the HBM peak (the figure that matters for the memory-bound ones)."""
    out = []
    for r in sorted(rows, key=lambda r: -r["ms"]):
        gbs = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0.0
        out.append({"kernel": r["kernel"], "launches": r["launches"] // calls, "ms": round(r["ms"] / calls, 4),
                    "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 3) if r["flops"] else None,
                    "gbs": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4)})
    return out


HBM_BOUND_KERNELS = ("stage_input", "wino_input", "wino_output", "proposal_decode", "radix_sort", "nms_mask", "nms_reduce",
                     "roipool", "split_k_finish", "bn_relu_avgpool")


def hbm_bound_stages(table):
    """The memory-bound kernels of a kernel table: algorithmic GB/s against the 8 TB/s HBM peak.  The sort and the
    NMS kernels are latency-bound at the default 1 000 candidates per frame (small launches); the dense-crowd
    record shows them at 97 920."""
    out = {}
    for r in table:
        if r["kernel"] in HBM_BOUND_KERNELS:
            out[r["kernel"]] = {"ms": r["ms"], "launches": r["launches"], "achieved": r["gbs"], "peak": PEAK_HBM_GBS,
                                "unit": "GB/s", "frac": r["hbm_frac"]}
    nms = [r for r in table if r["kernel"] in ("nms_mask", "nms_reduce")]
    if len(nms) == 2:
        ms = sum(r["ms"] for r in nms)
        gb = sum(r["gbs"] * r["ms"] for r in nms) / ms if ms > 0 else 0.0
        out["nms_mask+nms_reduce"] = {"ms": round(ms, 4), "achieved": round(gb, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": round(gb / PEAK_HBM_GBS, 4)}
    return out


def _timed(fn, reps, warm=2, sync_each=False):
    """Seconds per call of fn on the current stream (the host runs ahead of the GPU unless sync_each)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def _bracketed(fn, reps):
    from facedetection_amd import profiling
    profiling.begin()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return _kernel_table(profiling.end(), reps)


def config_records(det, frames, device):
    """The other configurations of the benchmark definition and the front-end resize, on this GPU, at the
    headline's precision (untimed with respect to the headline; rank 0 only).  `det` is the headline's
    detector, `frames` its resident 1080p frames."""
    from facedetection_amd import ops
    from facedetection_amd.symbol import processing as P
    from facedetection_amd.symbol.config import config
    from facedetection_amd.synthetic import crowd_image, synthetic_image
    rec = {}

    # configuration 2: one 1080p frame, batch 1, full cascade
    one = frames[:1]
    call = lambda: det.detect_batch_device(one)
    rec["latency_1080p"] = {
        "ms": round(1e3 * _timed(call, 30, sync_each=True), 3),
        "ms_back_to_back": round(1e3 * _timed(call, 30), 3), "unit": "ms per frame", "dtype": det.precision,
        "workload": "one 1080p frame resident in HBM through fd_detect, batch 1 (ms: host waits for every frame; "
                    "ms_back_to_back: calls queued without waiting)"}
    table = _bracketed(call, 10)
    rec["latency_1080p"]["kernel_launch_groups"] = int(sum(r["launches"] for r in table))
    rec["latency_1080p"]["kernels"] = table[:8]

    # configuration 4's frame size: 4K frames in micro-batches of 4 (their stage-1 tensors are as large as those
    # of 16 frames of 1080p)
    q = [synthetic_image(1080, 1920, seed=900 + i) for i in range(4)]
    k4 = np.concatenate([np.concatenate(q[:2], axis=1), np.concatenate(q[2:], axis=1)], axis=0)
    f4 = torch.from_numpy(np.stack([np.roll(k4, 97 * i, axis=1) for i in range(8)])).to(device)
    t = _timed(lambda: [det.detect_batch_device(f4[i:i + 4]) for i in (0, 4)], 3, warm=1)
    t1 = _timed(lambda: det.detect_batch_device(f4[:1]), 10, sync_each=True)
    rec["frame_4k"] = {"value": round(8 / t, 2), "unit": "frames/s per GPU", "ms_single_frame": round(1e3 * t1, 3),
                       "dtype": det.precision, "gflop_per_frame": round(2 * gmac_per_frame(2160, 3840), 1),
                       "workload": "8 synthetic 2160x3840 frames resident in HBM, micro-batches of 4, full cascade"}
    del f4

    # front-end resize of the command-line loop (detection.py:63-78, :110): 1080p -> 562 x 1000, 64 frames per launch
    n_rz = min(64, int(frames.shape[0]))
    scale = 1000.0 / 1920.0
    src = frames[:n_rz]
    small = ops.resize_bilinear(src, scale, scale)
    moved = src.numel() + small.numel()
    rows = [r for r in _bracketed(lambda: ops.resize_bilinear(src, scale, scale), 20) if r["kernel"] == "resize_linear_u8"]
    if rows:
        us = rows[0]["ms"] * 1e3 / max(1, rows[0]["launches"])
        gbs = moved / (us * 1e-6) / 1e9
        rec["frontend_resize"] = {
            "bound": "hbm", "kernel": "resize_linear_u8", "frames": n_rz, "us_per_launch": round(us, 2),
            "algorithmic_bytes": int(moved), "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(gbs / PEAK_HBM_GBS, 4),
            "workload": "%d frames of 1080x1920x3 uint8 -> %dx%d, OpenCV INTER_LINEAR fixed-point arithmetic; "
                        "bytes = 3 (H W + h' w') per frame" % (n_rz, small.shape[1], small.shape[2])}
    del small

    # configuration 5: dense crowd -- every anchor that passes the 12-pixel filter goes into the RPN NMS (no pre-NMS
    # cut), 6 000 rois into stage 4
    img = torch.from_numpy(crowd_image(1080, 1920, seed=2, faces=5000)[None]).to(device)
    saved = (config.TEST.RPN_MIN_SIZE, config.TEST.RPN_PRE_NMS_TOP_N, config.TEST.RPN_POST_NMS_TOP_N)
    config.TEST.RPN_MIN_SIZE, config.TEST.RPN_PRE_NMS_TOP_N, config.TEST.RPN_POST_NMS_TOP_N = 12, 0, 6000
    try:
        det.reconfigure()
        call = lambda: det.detect_batch_device(img)
        ms = 1e3 * _timed(call, 5, warm=2, sync_each=True)
        table = _bracketed(call, 3)
        dets, n, _ = call()
        rout = det.tensor("rpn_out")                                  # [1, fh, fw, 72] = bg(12) | fg(12) | deltas(48)
        fh, fw = rout.shape[1], rout.shape[2]
        base = P.generate_anchors(16, (0.5, 1, 2), (4, 8, 16, 32))
        sx, sy = np.meshgrid(np.arange(fw) * 16.0, np.arange(fh) * 16.0)
        shifts = np.stack([sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()], axis=1)
        anchors = (shifts[:, None, :] + base[None, :, :]).reshape(-1, 4).astype(np.float32)
        boxes = P.clip_boxes(P.bbox_pred(anchors, rout[0, :, :, 24:].reshape(-1, 4)), (1080, 1920))
        cand = int(((boxes[:, 2] - boxes[:, 0] + 1 >= 12) & (boxes[:, 3] - boxes[:, 1] + 1 >= 12)).sum())
        rec["dense_crowd"] = {
            "ms": round(ms, 3), "unit": "ms per frame", "dtype": det.precision, "candidates_into_rpn_nms": cand,
            "rois_into_stage4": 6000, "detections": int(n[0].item()),
            "gflop_per_frame": round(2 * gmac_per_frame(1080, 1920, 6000), 1),
            "workload": "one 1080p frame with 5 000 synthetic 12-40 px faces; RPN_MIN_SIZE 12, no pre-NMS cut, "
                        "RPN_POST_NMS_TOP_N 6000",
            "hbm_bound_stages": hbm_bound_stages(table), "kernels": table[:10]}
    finally:
        config.TEST.RPN_MIN_SIZE, config.TEST.RPN_PRE_NMS_TOP_N, config.TEST.RPN_POST_NMS_TOP_N = saved
        det.reconfigure()
    return rec


# ----------------------------------------------------------------------------- launching
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpu_count():
    """GPUs this process would see, counted without initialising HIP (torch.cuda.device_count() calls
    hipGetDeviceCount on ROCm): the visibility variables if set, else the compute nodes the kernel driver
    lists under /sys/class/kfd (GPU nodes are those with SIMDs)."""
    import glob
    nodes = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(prop) as f:
                fields = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            nodes += 1 if int(fields.get("simd_count", "0")) > 0 else 0
        except (OSError, ValueError):
            pass
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            nodes = min(nodes, len([x for x in v.split(",") if x.strip()]))
    return nodes


def under_profiler():
    """A profiler's preloaded library has initialised the GPU in this process before main() runs."""
    return ("rocprof" in os.environ.get("LD_PRELOAD", "").lower()
            or any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ))


def launch_ranks(args):
    """`--gpus N` without a launcher: start N ranks of this script, one per GPU.  Nothing in this
    process has touched a GPU (the devices are counted from the driver's topology files, not through
    HIP), and the ranks are fresh interpreters started as child processes.  Under a profiler the
    parent process is already initialised on the GPU, so the self-launcher refuses to run there:
    profile a single rank, or let torch.distributed.run start the ranks."""
    if not args.rehearse:
        if under_profiler():
            sys.stderr.write("bench.py: --gpus %d under a profiler: this process is already initialised on the GPU and "
                             "must not start further GPU processes; nothing was run\n" % args.gpus)
            return 2
        have = visible_gpu_count()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this host; nothing was run\n"
                             % (args.gpus, have))
            return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(args.gpus))
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def rehearsal_step(n_frames, rank, world, R=10):
    """Stand-in for the detector in --rehearse: deterministic per-frame detection lists."""
    dets = np.zeros((n_frames, R, 5), np.float32)
    counts = np.zeros((n_frames,), np.int32)
    for j in range(n_frames):
        g = j * world + rank
        counts[j] = (g * 3) % (R + 1)
        dets[j, :counts[j]] = np.random.default_rng(g).uniform(0, 500, (counts[j], 5)).astype(np.float32)
    return torch.from_numpy(dets), torch.from_numpy(counts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=256, help="frames per step per GPU")
    ap.add_argument("--batch", type=int, default=PMC_BATCH, help="frames per launch sequence")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams of the headline (float32) measurement; with one stream the per-launch "
                         "durations inside the timed region are those of an otherwise idle chip")
    ap.add_argument("--split-streams", type=int, default=2,
                    help="HIP streams of the f16x3 sub-record (its memory-bound and matrix-bound kernels overlap)")
    ap.add_argument("--two-streams", action="store_true",
                    help="also time the headline precision on two streams (sub-record two_streams; off by default so "
                         "that a kernel trace of the default run holds single-stream launches of the headline kernels only)")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3", "bf16x3"],
                    help="arithmetic of the HEADLINE: float32 operands (the reference's, default), or the f16x3 split")
    ap.add_argument("--height", type=int, default=1080, help="frame height of the headline workload")
    ap.add_argument("--width", type=int, default=1920, help="frame width of the headline workload")
    ap.add_argument("--split-record", action="store_true",
                    help="also time the opt-in f16x3 arithmetic (not the reference's) and report it as the sub-record f16x3")
    ap.add_argument("--no-split-record", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--no-b3-record", action="store_true",
                    help="skip the sub-record bf16x3 (float32 tensors, every product from three-way bf16 splits on the bf16 "
                         "matrix cores; its kernels carry their own names, so the float32 kernels' trace averages are unaffected)")
    ap.add_argument("--no-config-records", action="store_true",
                    help="skip the sub-records of the other configurations (single-frame latency, 4K, dense crowd, resize)")
    ap.add_argument("--check-frames", type=int, default=2, help="frames compared with the float64 oracle")
    ap.add_argument("--from-host", action="store_true",
                    help="frames start in pinned host memory and are uploaded inside the timed region "
                         "(PCIe-inclusive rate; the default keeps them resident in HBM as the metric is defined)")
    ap.add_argument("--direct-record", action="store_true",
                    help="also time the float32 pass with every 3x3 layer as a direct convolution (FD_WINOGRAD=0) and report "
                         "it as f32_direct_convolutions; off by default so that a kernel trace of the default run holds the "
                         "headline pass's launches of the float32 kernels only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--rehearse", action="store_true",
                    help="CPU rehearsal of launcher, barriers, timing and gather over gloo; no detector runs")
    args = ap.parse_args()

    global H, W, GFLOP_PER_FRAME
    H, W = args.height, args.width
    GFLOP_PER_FRAME = round(2 * gmac_per_frame(H, W, PROP["post_nms_top_n"]), 1)
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); they must agree\n" % (args.gpus, world))
        sys.exit(2)
    # under torch.distributed.run (RANK is set) the process group is used even for one rank, so the
    # launch path the driver takes for N > 1 can be rehearsed on a single GPU
    distributed = world > 1 or "RANK" in os.environ
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        # RCCL prints a version banner (and gloo its connection report) on the C-level stdout when the
        # communicator comes up; stdout carries the one JSON line, so they are sent to stderr (first
        # collective included)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.rehearse:
                dist.init_process_group(backend="gloo", rank=rank, world_size=world)
                dist.barrier()
            else:
                torch.cuda.set_device(local)        # every collective of this rank runs on its own GPU from the first one on
                dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local))
                dist.barrier()
                torch.cuda.synchronize()
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    def barrier():
        if distributed:
            dist.barrier()
        if not args.rehearse:
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if args.rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    from facedetection_amd.parallel import gather_detections

    if args.rehearse:
        # same protocol, no GPU: warm-up, barrier, K timed steps, barrier, max over ranks, gather
        for _ in range(args.warmup):
            rehearsal_step(args.frames, rank, world)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            dets, counts = rehearsal_step(args.frames, rank, world)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        gathered = gather_detections(dets, counts, rank, world, force_collective=distributed)
        seen = torch.tensor([1], dtype=torch.int64)
        if distributed:
            dist.all_reduce(seen)
        if rank == 0:
            expect = sum((g * 3) % 11 for g in range(args.frames * world))
            print(json.dumps({
                "metric": "1080p frames/sec (node)", "value": None, "unit": "frames/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": None,
                "data": "none (rehearsal: no detector ran, nothing was measured)", "rehearsal": True,
                "n_ranks_seen": int(seen.item()), "frames_gathered": len(gathered),
                "boxes_gathered": int(sum(len(g) for g in gathered)), "boxes_expected": int(expect),
                "config": {"workload": "launcher / barrier / gather rehearsal on CPU over gloo"}}), flush=True)
        if distributed:
            dist.destroy_process_group()
        return

    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from facedetection_amd import _lib, profiling
    from facedetection_amd.pipeline import FramePipeline

    do_split = args.split_record and not args.no_split_record and args.precision == "f32"
    n_streams = max(args.streams, args.split_streams if do_split else 1, 2 if args.two_streams else 1)
    pipe = FramePipeline("synthetic:0", local, streams=n_streams, precision=args.precision)
    frames = make_frames(args.frames, device, seed=rank)
    B = max(1, min(args.batch, args.frames))
    host_frames = frames.cpu().pin_memory() if args.from_host else None

    def measure(precision, streams):
        """W warm-up steps, one untimed single-stream step with every kernel group bracketed by HIP
        events (tells which kernel dominates; its rows are the per-kernel table), then EXACTLY K timed
        steps between barriers with only the dominant kernel bracketed (two event records around
        each of the other ~110 launches per micro-batch cost 1-4 % of the frame rate)."""
        pipe.set_precision(precision)

        def step():
            if host_frames is None:
                return pipe.detect_device(frames, B, streams=streams)
            outs = []
            for i in range(0, args.frames, 4 * B):        # upload a few micro-batches at a time
                chunk = host_frames[i:i + 4 * B].to(device, non_blocking=True)
                outs += pipe.detect_device(chunk, B, streams=streams)
            return outs

        for _ in range(args.warmup):
            step()
        barrier()
        iso_rows, dom_name = [], None
        if not args.no_profile:
            profiling.begin()
            for i in range(0, args.frames, B):
                pipe.detectors[0].detect_batch_device(frames[i:i + B])
            torch.cuda.synchronize()
            iso_rows = profiling.end()
            conv = [r for r in iso_rows if r["kernel"].startswith("conv_")]
            dom_name = max(conv, key=lambda r: r["ms"])["kernel"] if conv else None
            barrier()
        if dom_name:
            profiling.begin(only=dom_name)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            outs = step()
        barrier()
        elapsed = time.perf_counter() - t0
        rows = profiling.end() if dom_name else []
        return dict(elapsed=max_over_ranks(elapsed), outs=outs, rows=rows, iso_rows=iso_rows, streams=streams,
                    precision=precision)

    def roofline_of(m, fps):
        rows, iso_rows = m["rows"], m["iso_rows"]
        if not rows:
            return None, None
        dom = rows[0]
        conv = [r for r in iso_rows if r["kernel"].startswith("conv_")]
        tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        # split precision executes 3 f16 MFMA FLOPs per algorithmic FLOP, so the ceiling for
        # ALGORITHMIC FLOP/s on that path is the dense f16 MFMA peak / 3
        split = dom["kernel"].startswith("conv_split")
        b3 = "_b3" in dom["kernel"]
        peak = PEAK_F16_MATRIX_TFLOPS / 3.0 if split else PEAK_F16_MATRIX_TFLOPS / 6.0 if b3 else PEAK_FP32_MATRIX_TFLOPS
        traffic, note = pmc_traffic(dom["kernel"]) if B == PMC_BATCH else (None, "PMC passes are taken at micro-batch %d" % PMC_BATCH)
        roof = {
            "bound": "mfma", "kernel": dom["kernel"], "achieved": round(tf, 3), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": traffic,
            "traffic_note": note + ("; algorithmic bytes per launch %.0f" % (dom["bytes"] / dom["launches"])),
            "peak_note": ("dense f16 MFMA peak 2500 TFLOP/s / 3 MFMAs per product" if split else
                          "dense bf16 MFMA peak 2500 TFLOP/s / 6 MFMAs per product" if b3 else "dense fp32 MFMA peak"),
            "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
            "flop_per_launch": round(dom["flops"] / dom["launches"]),
            "measured": "HIP events around every launch of this kernel inside the timed region (%d stream%s)"
                        % (m["streams"], "" if m["streams"] == 1 else "s: launches overlap the other stream's kernels"),
        }
        # The wide stride-1 3x3 convolutions (RPN, 256-channel stage; float32 also the 128-channel stage) run in
        # Winograd F(4x4, 3x3) form (1/4 of the multiplies).  Kernel rows and `achieved` count the FLOPs a kernel
        # EXECUTES, and so does the pass's utilisation (pipeline_executed_*); the graph definition's FLOPs (every
        # layer direct: the metric's 354.5 GFLOP per 1080p frame) give an algorithmic rate reported beside it.
        executed = sum(r["flops"] for r in iso_rows) / args.frames / 1e9
        roof["executed_gflop_per_frame"] = round(executed, 2)
        roof["pipeline_executed_tflops"] = round(executed * fps / 1e3, 3)
        roof["pipeline_executed_frac"] = round(executed * fps / 1e3 / peak, 4)
        roof["direct_equivalent_gflop_per_frame"] = GFLOP_PER_FRAME
        roof["pipeline_direct_equivalent_tflops"] = round(GFLOP_PER_FRAME * fps / 1e3, 3)
        roof["flop_note"] = ("pipeline_executed_* = FLOPs the kernels executed over the whole pass x frames/s against the "
                             "matrix peak: the utilisation of the pass (Winograd F(4x4,3x3) layers count a quarter of their "
                             "direct form, GEMM row padding included).  pipeline_direct_equivalent_tflops = %.1f GFLOP per "
                             "frame (every layer as a direct convolution, the graph's definition) x frames/s: an algorithmic "
                             "rate, NOT a utilisation -- part of those FLOPs were never executed" % GFLOP_PER_FRAME)
        iso = [r for r in iso_rows if r["kernel"] == dom["kernel"]][0]
        itf = iso["flops"] / (iso["ms"] * 1e-3) / 1e12
        roof["isolated"] = {
            "achieved": round(itf, 3), "frac": round(itf / peak, 4),
            "avg_launch_ms": round(iso["ms"] / iso["launches"], 4),
            "all_conv_tflops": round(sum(r["flops"] for r in conv) / (sum(r["ms"] for r in conv) * 1e-3) / 1e12, 3),
            "note": "same kernel in one extra untimed step on a single stream"}
        return roof, _kernel_table(iso_rows)

    def measure_direct():
        """The float32 pass with every 3x3 layer in its direct form (FD_WINOGRAD=0; the form is chosen when a
        detector is created, hence a second detector): one warm-up step, then K timed steps between barriers."""
        old = os.environ.get("FD_WINOGRAD")
        os.environ["FD_WINOGRAD"] = "0"
        try:
            direct = FramePipeline("synthetic:0", local, streams=1, precision="f32")
        finally:
            if old is None:
                del os.environ["FD_WINOGRAD"]
            else:
                os.environ["FD_WINOGRAD"] = old
        try:
            direct.detect_device(frames, B, streams=1)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                direct.detect_device(frames, B, streams=1)
            barrier()
            return max_over_ranks(time.perf_counter() - t0)
        finally:
            direct.close()

    head = measure(args.precision, args.streams)
    # the same pass with micro-batches alternating over two streams (reported beside the headline: kernels
    # of the two streams overlap, so per-kernel durations there say nothing about a kernel's own roofline)
    head2 = measure(args.precision, 2) if (args.two_streams and args.streams == 1 and n_streams >= 2) else None
    sub = measure("f16x3", args.split_streams) if do_split else None
    do_b3 = args.precision == "f32" and not args.no_b3_record and (H, W) == (1080, 1920)
    sub3 = measure("bf16x3", 1) if do_b3 else None
    direct_elapsed = measure_direct() if (args.direct_record and args.precision == "f32") else None

    # hand-off (untimed): per-frame detection lists gathered in global frame order
    outs = head["outs"]
    dets = torch.cat([o[0] for o in outs])
    counts = torch.cat([o[1] for o in outs])
    gathered = gather_detections(dets, counts, rank, world, force_collective=distributed)
    n_boxes = int(sum(len(g) for g in gathered)) if gathered is not None else 0
    seen = torch.tensor([1], dtype=torch.int64, device=device)
    if distributed:
        dist.all_reduce(seen)
        n_seen = int(seen.item())
        # Everything below is rank 0's alone (per-GPU sub-records, the float64 check, the CPU baseline: minutes of
        # host work).  The process group ends here on every rank, so no rank sits in a collective with a watchdog
        # running, and the other ranks leave the host's cores to the baseline.
        barrier()
        dist.destroy_process_group()
    else:
        n_seen = int(seen.item())
    if rank != 0:
        pipe.close()
        return

    total_frames = args.frames * args.steps * world
    fps = total_frames / head["elapsed"]
    split_head = args.precision == "f16x3"
    line = {
        "metric": "1080p frames/sec (node)" if (H, W) == (1080, 1920) else "%dx%d frames/sec (node)" % (H, W),
        "value": round(fps, 3),
        "unit": "frames/s",
        "n_gpus": world,
        "n_ranks_seen": n_seen,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1000.0 * head["elapsed"] / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("f32 operands as fp16 hi+lo pairs (22 significant bits, fp16 range), 3 f16 MFMAs per product, f32 accumulate"
                  if split_head else B3_DTYPE if args.precision == "bf16x3" else "f32"),
        "data": "synthetic",
        "config": {"workload": "batch of %d synthetic %s frames per GPU resident in HBM, full Faster R-CNN "
                               "(ResNet-50) cascade, micro-batches of %d on %d HIP stream(s)"
                               % (args.frames, "1080p" if (H, W) == (1080, 1920) else "%dx%d" % (H, W), B, args.streams),
                   "frames_per_step_per_gpu": args.frames, "micro_batch": B, "streams": args.streams,
                   "height": H, "width": W,
                   "rpn_pre_nms_top_n": 1000, "rpn_post_nms_top_n": 10, "rpn_min_size": 50,
                   "weights": "seeded random (synthetic:0)", "detections_in_last_step": n_boxes,
                   "convolution_forms": (("the RPN's 3x3 layer and the stride-1 3x3 layers of the 256-channel stage run as "
                                          "Winograd F(4x4,3x3) on fp16 hi+lo pairs (FD_WINOGRAD=%s), the rest direct"
                                          if split_head else
                                          "every layer a float32 convolution of the reference's graph; the stride-1 3x3 "
                                          "layers with 128 or more channels and the RPN's 3x3 run as Winograd "
                                          "F(4x4,3x3) in float32 (FD_WINOGRAD=%s), the rest direct; the measured error "
                                          "against float64 is in accuracy_vs_float64")
                                         % os.environ.get("FD_WINOGRAD", "4")),
                   "frames_start": "pinned host memory (PCIe upload timed)" if args.from_host else "HBM",
                   "library_build": _lib.lib().fd_build_id().decode("ascii")},
    }
    roof, table = roofline_of(head, fps)
    if roof:
        line["roofline"] = roof
        line["hbm_bound_stages"] = hbm_bound_stages(table)
        line["kernels_single_stream_step"] = table
    if not split_head and B == PMC_BATCH and (H, W) == (1080, 1920):
        line["hbm_whole_pass"] = hbm_whole_pass(fps)
    if head2:
        line["two_streams"] = {"value": round(total_frames / head2["elapsed"], 3), "unit": "frames/s",
                               "ms_per_step": round(1000.0 * head2["elapsed"] / args.steps, 3), "streams": 2,
                               "dtype": line["dtype"]}
    if direct_elapsed:
        line["f32_direct_convolutions"] = {
            "value": round(total_frames / direct_elapsed, 3), "unit": "frames/s",
            "ms_per_step": round(1000.0 * direct_elapsed / args.steps, 3), "streams": 1, "dtype": "f32",
            "pipeline_executed_tflops": round(GFLOP_PER_FRAME * total_frames / direct_elapsed / 1e3, 3),
            "pipeline_executed_frac": round(GFLOP_PER_FRAME * total_frames / direct_elapsed / 1e3 / PEAK_FP32_MATRIX_TFLOPS, 4),
            "note": "the headline pass with FD_WINOGRAD=0: every 3x3 layer as a direct float32 convolution "
                    "(%.1f GFLOP executed per frame), for readers who want the headline without the Winograd form"
                    % GFLOP_PER_FRAME}
    acc = None
    if args.check_frames > 0 and not split_head:
        acc = accuracy_vs_float64(pipe, frames, min(args.check_frames, args.frames),
                                  (args.precision,) + (("f16x3",) if do_split else ()) + (("bf16x3",) if sub3 else ()))
        line["accuracy_vs_float64"] = acc[args.precision]
    if sub:
        sfps = total_frames / sub["elapsed"]
        sroof, stable = roofline_of(sub, sfps)
        clamped_timed = int(sum(int(o[2][1].item()) for o in sub["outs"]))
        rec = {"value": round(sfps, 3), "unit": "frames/s", "ms_per_step": round(1000.0 * sub["elapsed"] / args.steps, 3),
               "streams": args.split_streams,
               "dtype": "f32 operands as fp16 hi+lo pairs (22 significant bits, fp16 range), 3 f16 MFMAs per "
                        "product, f32 accumulate",
               "roofline": sroof, "kernels_single_stream_step": stable,
               "clamped_values_last_timed_step": clamped_timed}
        if acc:
            rec.update(acc["f16x3"])
        line["f16x3"] = rec
    if sub3:
        b3fps = total_frames / sub3["elapsed"]
        b3roof, b3table = roofline_of(sub3, b3fps)
        rec = {"value": round(b3fps, 3), "unit": "frames/s", "ms_per_step": round(1000.0 * sub3["elapsed"] / args.steps, 3),
               "streams": 1, "dtype": B3_DTYPE, "roofline": b3roof, "kernels_single_stream_step": (b3table or [])[:12],
               "note": "opt-in arithmetic (precision=\"bf16x3\"), reported beside the float32 headline, never as it"}
        if acc:
            rec.update(acc["bf16x3"])
        line["bf16x3"] = rec
    if not args.no_config_records and not split_head and (H, W) == (1080, 1920):
        pipe.set_precision(args.precision)
        line["configs"] = config_records(pipe.detectors[0], frames, device)
    pipe.close()
    del frames
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline()
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
