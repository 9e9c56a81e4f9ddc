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
matrix cores -- the reference's arithmetic.  The same run then times the library's default
arithmetic (f16x3: every float32 operand as an exact-to-22-bit fp16 hi+lo pair, three f16 MFMAs
per product) and reports it as the sub-record `f16x3`, together with its measured error against
the float64 oracle on frames of this benchmark and the number of values it had to clamp.

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

H, W = 1080, 1920
GFLOP_PER_FRAME = 354.5          # algorithmic, post_nms_top_n = 10 (2 x 177.25 GMAC)
PEAK_FP32_MATRIX_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MATRIX_TFLOPS = 2500.0  # v_mfma_f32_16x16x32_f16, dense
PEAK_HBM_GBS = 8000.0            # HBM3E peak
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
    m = re.match(r"conv_(split|f32|igemm)_(\d+)x(\d+)(_pre|_o\d)?$", kernel)
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
    elif pre:       # BatchNorm-on-load kernel: <BM, BN, mode 0, float32 operands?>
        want = re.compile(r"conv_split_reg_kernel<%s, %s, 0, %s>" % (bm, bn, "true" if family == "f32" else "false"))
    else:           # direct-to-LDS kernel: <BM, BN, stages, waves, float32 operands?, workgroups per CU>
        want = re.compile(r"conv_split_glds_kernel<%s, %s, \d+, \d+, %s, %s>" % (bm, bn, "true" if family == "f32" else "false", occ))
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


def accuracy_vs_float64(pipe, frames, n_check):
    """Untimed: the graph outputs of the first `n_check` benchmark frames in both precisions against
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
        for prec in ("f32", "f16x3"):
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


# ----------------------------------------------------------------------------- launching
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """`--gpus N` without a launcher: start N ranks of this script, one per GPU.  Nothing in this
    process has touched a GPU (device_count does not initialise the runtime), and the ranks are
    fresh interpreters, not forks or re-execs of an initialised process."""
    if not args.rehearse:
        have = torch.cuda.device_count()
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
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3"],
                    help="arithmetic of the HEADLINE: float32 operands (the reference's, default), or the f16x3 split")
    ap.add_argument("--no-split-record", action="store_true", help="skip the f16x3 sub-record and the accuracy check")
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

    do_split = not args.no_split_record and args.precision == "f32"
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
        peak = PEAK_F16_MATRIX_TFLOPS / 3.0 if split else PEAK_FP32_MATRIX_TFLOPS
        traffic, note = pmc_traffic(dom["kernel"]) if B == PMC_BATCH else (None, "PMC passes are taken at micro-batch %d" % PMC_BATCH)
        roof = {
            "bound": "mfma", "kernel": dom["kernel"], "achieved": round(tf, 3), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": traffic,
            "traffic_note": note + ("; algorithmic bytes per launch %.0f" % (dom["bytes"] / dom["launches"])),
            "peak_note": ("dense f16 MFMA peak 2500 TFLOP/s / 3 MFMAs per product" if split else "dense fp32 MFMA peak"),
            "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
            "flop_per_launch": round(dom["flops"] / dom["launches"]),
            "measured": "HIP events around every launch of this kernel inside the timed region (%d stream%s)"
                        % (m["streams"], "" if m["streams"] == 1 else "s: launches overlap the other stream's kernels"),
            "pipeline_tflops": round(GFLOP_PER_FRAME * fps / 1e3, 3),
            "pipeline_frac": round(GFLOP_PER_FRAME * fps / 1e3 / peak, 4),
        }
        # Both passes run the wide stride-1 3x3 convolutions (RPN, 256-channel stage; float32 also the
        # 128-channel stage) in Winograd F(4x4, 3x3) form (1/4 of the multiplies).  Kernel rows and `achieved` count the FLOPs a kernel
        # EXECUTES; pipeline_tflops counts the layer definitions' FLOPs (the direct form, the figure the
        # metric's 354.5 GFLOP per frame stands for).  The executed total is reported next to it.
        executed = sum(r["flops"] for r in iso_rows) / args.frames / 1e9
        roof["executed_gflop_per_frame"] = round(executed, 2)
        roof["pipeline_executed_tflops"] = round(executed * fps / 1e3, 3)
        roof["pipeline_executed_frac"] = round(executed * fps / 1e3 / peak, 4)
        roof["flop_note"] = ("pipeline_tflops = %.1f GFLOP per frame (every layer as a direct convolution) x frames/s; "
                             "pipeline_executed_* = FLOPs the kernels executed (Winograd F(4x4,3x3) layers at 1/4 of "
                             "the direct form's, GEMM row padding included)" % GFLOP_PER_FRAME)
        iso = [r for r in iso_rows if r["kernel"] == dom["kernel"]][0]
        itf = iso["flops"] / (iso["ms"] * 1e-3) / 1e12
        roof["isolated"] = {
            "achieved": round(itf, 3), "frac": round(itf / peak, 4),
            "avg_launch_ms": round(iso["ms"] / iso["launches"], 4),
            "all_conv_tflops": round(sum(r["flops"] for r in conv) / (sum(r["ms"] for r in conv) * 1e-3) / 1e12, 3),
            "note": "same kernel in one extra untimed step on a single stream"}
        table = [{"kernel": r["kernel"], "launches": r["launches"], "ms": round(r["ms"], 3),
                  "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 3) if r["flops"] else None,
                  "gbs": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)} for r in sorted(iso_rows, key=lambda r: -r["ms"])]
        return roof, table

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

    if rank == 0:
        total_frames = args.frames * args.steps * world
        fps = total_frames / head["elapsed"]
        split_head = args.precision == "f16x3"
        line = {
            "metric": "1080p frames/sec (node)",
            "value": round(fps, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "n_ranks_seen": int(seen.item()),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1000.0 * head["elapsed"] / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 operands as fp16 hi+lo pairs (22 significant bits, fp16 range), 3 f16 MFMAs per product, f32 accumulate"
                      if split_head else "f32"),
            "data": "synthetic",
            "config": {"workload": "batch of %d synthetic 1080p frames per GPU resident in HBM, full Faster R-CNN "
                                   "(ResNet-50) cascade, micro-batches of %d on %d HIP stream(s)"
                                   % (args.frames, B, args.streams),
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
            line["kernels_single_stream_step"] = table
        if not split_head and B == PMC_BATCH:
            line["hbm_whole_pass"] = hbm_whole_pass(fps)
        if head2:
            line["two_streams"] = {"value": round(total_frames / head2["elapsed"], 3), "unit": "frames/s",
                                   "ms_per_step": round(1000.0 * head2["elapsed"] / args.steps, 3), "streams": 2,
                                   "dtype": line["dtype"]}
        if direct_elapsed:
            line["f32_direct_convolutions"] = {
                "value": round(total_frames / direct_elapsed, 3), "unit": "frames/s",
                "ms_per_step": round(1000.0 * direct_elapsed / args.steps, 3), "streams": 1, "dtype": "f32",
                "pipeline_tflops": round(GFLOP_PER_FRAME * total_frames / direct_elapsed / 1e3, 3),
                "pipeline_frac": round(GFLOP_PER_FRAME * total_frames / direct_elapsed / 1e3 / PEAK_FP32_MATRIX_TFLOPS, 4),
                "note": "the headline pass with FD_WINOGRAD=0: every 3x3 layer as a direct float32 convolution "
                        "(%.1f GFLOP executed per frame: pipeline_tflops are executed FLOPs here), for readers who want the "
                        "headline without the Winograd form"
                        % GFLOP_PER_FRAME}
        acc = None
        if do_split and args.check_frames > 0:
            acc = accuracy_vs_float64(pipe, frames, min(args.check_frames, args.frames))
            line["accuracy_vs_float64"] = acc["f32"]
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
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if distributed:
        barrier()              # the other ranks wait here while rank 0 runs the untimed checks
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
