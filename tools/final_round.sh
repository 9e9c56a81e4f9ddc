#!/bin/bash
# Everything profiles/ holds for one kernel set, in one GPU job:  bash tools/final_round.sh <tag>
# (profile_round.sh twice: the second pass finds the first pass's PMC file, stamped with the running
# library's source hash, and so quotes roofline.traffic in its bench lines)
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
bash tools/profile_round.sh $TAG
cp $OUT/pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
bash tools/profile_round.sh $TAG
timeout -k 10 200 python3 tools/frontend_bench.py 64 > $OUT/frontend.json 2>/dev/null
timeout -k 10 200 python3 tools/frontend_bench.py 128 > $OUT/frontend_128.json 2>/dev/null
timeout -k 10 300 python3 tools/time_configs.py f16x3 > $OUT/cfg_split.txt 2>&1
timeout -k 10 300 python3 tools/time_configs.py f32 > $OUT/cfg_f32.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt2 -- python3 $R/bench.py --frames 16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --check-frames 0 --no-split-record > $OUT/b16.json 2> $OUT/b16.err
cd $R
python3 tools/conv_layer_report.py $(ls $OUT/kt2/*/*kernel_trace.csv | head -1) 16 f32 > $OUT/layers_f32.txt 2>&1 || true
rm -rf $OUT/kt2
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt3 -- python3 $R/bench.py --precision f16x3 --streams 1 --frames 16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --check-frames 0 --no-split-record > $OUT/b16s.json 2> $OUT/b16s.err
cd $R
python3 tools/conv_layer_report.py $(ls $OUT/kt3/*/*kernel_trace.csv | head -1) 16 f16x3 > $OUT/layers_f16x3.txt 2>&1 || true
rm -rf $OUT/kt3
echo final round done
