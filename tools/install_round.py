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
for name in ("kernel_stats.csv", "bench.json", "bench_under_rocprof.json", "pmc_traffic.json", "bench_two_streams.json",
             "frontend.json", "frontend_128.json", "bench_direct.json"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, name)))
for name in ("layers_f32.txt", "layers_f16x3.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, name)))
for a, b in (("cfg_split.txt", "configs_f16x3.json"), ("cfg_f32.txt", "configs_f32.json")):
    line = [l for l in open(os.path.join(src, a)) if l.startswith("{")][-1]
    open(os.path.join(dst, "%s_%s" % (tag, b)), "w").write(line)
for name in ("bench.json", "bench_under_rocprof.json"):
    d = json.load(open(os.path.join(dst, "%s_%s" % (tag, name))))
    r = d["roofline"]
    print(name, d["value"], "fps;", r["kernel"], r["achieved"], "TF frac", r["frac"], "traffic", r["traffic"], "pipeline",
          r["pipeline_tflops"], r["pipeline_frac"], "executed", r["pipeline_executed_tflops"], r["pipeline_executed_frac"],
          "avg launch ms", r["avg_launch_ms"], "launches", r["launches"], "build", d["config"]["library_build"])
    print("  accuracy", d.get("accuracy_vs_float64"), "| f16x3", d["f16x3"]["value"], d["f16x3"]["roofline"]["frac"],
          d["f16x3"]["roofline"]["isolated"]["frac"], "| cpu", d["cpu_baseline"]["value"])
d = json.load(open(os.path.join(dst, "%s_bench_two_streams.json" % tag)))
print("two streams", d["value"], d["two_streams"]["value"])
d = json.load(open(os.path.join(dst, "%s_bench_direct.json" % tag)))
print("direct convolutions", d["value"], d["f32_direct_convolutions"])
print(open(os.path.join(dst, "%s_frontend.json" % tag)).read()[:800])
print(open(os.path.join(dst, "%s_frontend_128.json" % tag)).read()[:500])
for b in ("configs_f16x3.json", "configs_f32.json"):
    d = json.load(open(os.path.join(dst, "%s_%s" % (tag, b))))
    print(b, {k: (v["ms"], v.get("ms_without_event_brackets"), v.get("ms_recorded_pass")) for k, v in d.items() if isinstance(v, dict)})
