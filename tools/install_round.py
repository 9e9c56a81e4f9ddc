#!/usr/bin/env python3
"""Copies the files of gpurun_out/<tag>/ that are kept under profiles/ (run here after tools/final_round.sh
ran on the GPU box) and prints the figures the documents quote:  python tools/install_round.py <tag>"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for name in ("kernel_stats.csv", "kernel_stats_headline.csv", "bench_headline_under_rocprof.json", "bench.json", "bench_under_rocprof.json", "pmc_traffic.json", "bench_two_streams.json",
             "frontend.json", "frontend_128.json", "bench_direct.json", "bench_from_host.json", "frontend_pmc_traffic.json", "pmc_traffic_f32.json"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, name)))
for name in ("layers_f32.txt", "layers_f16x3.txt", "ridge_f32.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, name)))
# SQ-counter table of the float32 pass: header with the command, then the table tools/sq_summary.py printed on the box
if os.path.exists(os.path.join(src, "sq_table.md")):
    build = json.load(open(os.path.join(src, "bench.json")))["config"]["library_build"]
    with open(os.path.join(dst, "%s_sq_counters_f32.md" % tag), "w") as f:
        f.write("# SQ counters of one float32 pass (16 frames of 1080p, micro-batch 16, warm-up + one step), library build %s\n\n" % build)
        f.write("Command (one `--pmc` pass, counters only; durations come from separate `--kernel-trace --stats` runs):\n"
                "`rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS "
                "SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py --frames 16 --steps 1 --warmup 1 --no-cpu-baseline "
                "--no-profile --check-frames 0 --no-split-record`,\nsummarised by `tools/sq_summary.py` (kernel families summed over their "
                "launches; the detector's set-up kernels and torch's\nframe-generation kernels are in the list as well).\n\n"
                "Columns: matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1 024 SIMDs x GRBM_GUI_ACTIVE / 8); clock = GRBM_GUI_ACTIVE / 8 /\n"
                "dispatch duration (meaningful for the long kernels only: the counter window of a short dispatch is longer than the "
                "dispatch);\nwaves waiting = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.\n\n")
        f.write(open(os.path.join(src, "sq_table.md")).read())
        f.write("\nReading: the float32 GEMM families and the fused stem keep the fp32 matrix pipes busy 72-75 % of their launches at a\n"
                "sustained 2.2-2.4 GHz (the f16 kernels of round 1 ran at 1.8 GHz); `roofline.frac` of the dominant kernel is that figure\n"
                "times the clock ratio.  LDS waits are 1-2 % of wave time: operands arrive by LDS-DMA and fragment reads are conflict-free;\n"
                "what the waves wait for is the matrix pipe itself (one wave per SIMD and workgroup, two or three workgroups per CU) and,\n"
                "outside the K loop, the tile's epilogue traffic.  The short-K 1x1 layer on `<128,64,3,4,true,0>` (K = 64) shows what is\n"
                "left when the K loop is two steps long.\n")
for a, b in (("cfg_split.txt", "configs_f16x3.json"), ("cfg_f32.txt", "configs_f32.json")):
    line = [l for l in open(os.path.join(src, a)) if l.startswith("{")][-1]
    open(os.path.join(dst, "%s_%s" % (tag, b)), "w").write(line)
for name in ("bench.json", "bench_under_rocprof.json"):
    d = json.load(open(os.path.join(dst, "%s_%s" % (tag, name))))
    r = d["roofline"]
    print(name, d["value"], "fps;", r["kernel"], r["achieved"], "TF frac", r["frac"], "traffic", r["traffic"], "direct-equivalent",
          r["pipeline_direct_equivalent_tflops"], "executed", r["pipeline_executed_tflops"], r["pipeline_executed_frac"],
          "avg launch ms", r["avg_launch_ms"], "launches", r["launches"], "build", d["config"]["library_build"])
    print("  accuracy", d.get("accuracy_vs_float64"), "| bf16x3", d["bf16x3"]["value"], d["bf16x3"]["roofline"]["frac"],
          d["bf16x3"]["max_abs_score_err_vs_f64"], "| cpu", d["cpu_baseline"]["value"])
    c = d["configs"]
    print("  configs: latency", c["latency_1080p"]["ms"], c["latency_1080p"]["ms_back_to_back"], "| 4K", c["frame_4k"]["value"],
          c["frame_4k"]["ms_single_frame"], "| crowd", c["dense_crowd"]["ms"], "| resize", c["frontend_resize"]["us_per_launch"],
          c["frontend_resize"]["frac"])
d = json.load(open(os.path.join(dst, "%s_bench_two_streams.json" % tag)))
print("two streams", d["value"], d["two_streams"]["value"])
d = json.load(open(os.path.join(dst, "%s_bench_headline_under_rocprof.json" % tag)))
print("headline alone under rocprof", d["value"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"])
print("hbm_whole_pass", json.load(open(os.path.join(dst, "%s_bench.json" % tag))).get("hbm_whole_pass"))
d = json.load(open(os.path.join(dst, "%s_bench_from_host.json" % tag)))
print("from host", d["value"], d["config"]["library_build"])
d = json.load(open(os.path.join(dst, "%s_frontend_pmc_traffic.json" % tag)))
print("resize traffic", {k: (v["fetch_bytes_per_launch"], v["write_bytes_per_launch"]) for k, v in d["kernels"].items() if "resize3" in k})
d = json.load(open(os.path.join(dst, "%s_bench_direct.json" % tag)))
print("direct convolutions", d["value"], d["f32_direct_convolutions"])
print(open(os.path.join(dst, "%s_frontend.json" % tag)).read()[:800])
print(open(os.path.join(dst, "%s_frontend_128.json" % tag)).read()[:500])
for b in ("configs_f16x3.json", "configs_f32.json"):
    d = json.load(open(os.path.join(dst, "%s_%s" % (tag, b))))
    print(b, {k: (v["ms"], v.get("ms_without_event_brackets"), v.get("ms_recorded_pass")) for k, v in d.items() if isinstance(v, dict)})
