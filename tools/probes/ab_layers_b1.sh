#!/bin/bash
# per-layer table of a single frame for two settings
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
    name=${spec%%:*}; vars=${spec#*:}
    ( IFS=,; for v in $vars; do [ -n "$v" ] && export "$v"; done
      timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$name -- python3 $R/bench.py --frames 1 --batch 1 --steps 3 --warmup 2 --no-cpu-baseline --no-profile --check-frames 0 --no-config-records --no-b3-record > $OUT/ab_$name.json 2> $OUT/ab_$name.err )
    python3 $R/tools/conv_layer_report.py $(ls $OUT/kt_$name/*/*kernel_trace.csv | head -1) 1 f32 > $OUT/b1_layers_$name.txt
    rm -rf $OUT/kt_$name
    tail -11 $OUT/b1_layers_$name.txt | head -2
done
