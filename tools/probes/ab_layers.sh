#!/bin/bash
# Per-layer table of one micro-batch of 16 frames for each environment setting given as NAME:VAR=VALUE[,VAR=VALUE]
#   bash tools/probes/ab_layers.sh <tag> base: a:FD_X=1 b:FD_X=1,FD_Y=2
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
    name=${spec%%:*}; vars=${spec#*:}
    ( IFS=,; for v in $vars; do [ -n "$v" ] && export "$v"; done
      timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$name -- python3 $R/bench.py --frames 16 --batch 16 --steps 3 --warmup 2 --no-cpu-baseline --no-profile --check-frames 0 --no-config-records --no-b3-record > $OUT/ab_$name.json 2> $OUT/ab_$name.err
      timeout -k 10 150 python3 $R/bench.py --frames 128 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --check-frames 0 --no-config-records --no-b3-record > $OUT/fps_$name.json 2>> $OUT/ab_$name.err )
    python3 $R/tools/conv_layer_report.py $(ls $OUT/kt_$name/*/*kernel_trace.csv | head -1) 16 f32 > $OUT/ab_layers_$name.txt
    rm -rf $OUT/kt_$name
    echo $name $(python3 -c "import json;print(json.load(open('$OUT/fps_$name.json'))['value'])") $(tail -12 $OUT/ab_layers_$name.txt | head -1)
done
