#!/bin/bash
# Per-layer times of a single 1080p frame (batch 1) with one tile shape forced per kernel family
# (FD_F32_TILE, FD_F32_TILE_PRE, FD_F32_GEMM_TILE):  bash tools/probes/tile_sweep_b1.sh <tag> [batch]
set -e
TAG=$1
B=${2:-1}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
    name=$1
    timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$name -- python3 $R/bench.py --frames $B --batch $B --steps 3 --warmup 2 --no-cpu-baseline --no-profile --check-frames 0 --no-config-records --no-b3-record > $OUT/sweep_$name.json 2> $OUT/sweep_$name.err
    python3 $R/tools/conv_layer_report.py $(ls $OUT/kt_$name/*/*kernel_trace.csv | head -1) $B f32 > $OUT/sweep_layers_$name.txt
    rm -rf $OUT/kt_$name
}
run base
for t in 1 2 3 4; do export FD_F32_TILE=$t; run tile$t; done
unset FD_F32_TILE
for t in 1 2 3; do export FD_F32_TILE_PRE=$t; run pre$t; done
unset FD_F32_TILE_PRE
for t in 1 2 3 4; do export FD_F32_GEMM_TILE=$t; run gemm$t; done
echo sweep done
