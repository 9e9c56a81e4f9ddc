#!/bin/bash
# The default benchmark line and a short summary of it:  bash tools/quick_bench.sh <tag> [bench.py arguments]
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 500 python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
python3 - $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(d["value"], "fps;", r["kernel"], r["achieved"], "TF frac", r["frac"], "executed frac", r["pipeline_executed_frac"], "build", d["config"]["library_build"])
for k in d["kernels_single_stream_step"][:10]:
    print("  ", k)
if "bf16x3" in d:
    print("bf16x3", d["bf16x3"]["value"], d["bf16x3"]["roofline"]["frac"])
c = d.get("configs")
if c:
    print("latency", c["latency_1080p"]["ms"], c["latency_1080p"]["ms_back_to_back"], "4k", c["frame_4k"]["value"], c["frame_4k"]["ms_single_frame"],
          "crowd", c["dense_crowd"]["ms"], "resize", c["frontend_resize"]["frac"])
print("accuracy", d.get("accuracy_vs_float64"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
