#!/bin/bash
# Kernel statistics and HBM traffic of the benchmark for profiles/ (run on the GPU box):
#   bash tools/profile_round.sh <tag>      e.g. r02_b
# 1. rocprofv3 --kernel-trace --stats of the default `python3 bench.py`, and of the headline measurement alone
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) of a short run at the same
#    micro-batch, summarised by tools/pmc_summary.py with the library's source hash.
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cp $(ls $OUT/kt/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/kt
# the headline alone (no sub-records of the other configurations, no bf16x3 record): every launch of the float32
# kernels in this trace belongs to the 1080p micro-batches of 16, so its per-kernel averages are the headline's
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kth -- python3 $R/bench.py --no-config-records --no-b3-record --no-cpu-baseline > $OUT/bench_headline_under_rocprof.json 2> $OUT/kth.err
cp $(ls $OUT/kth/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_headline.csv
rm -rf $OUT/kth
SHORT="--frames 16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --check-frames 0 --split-streams 1 --no-b3-record --no-config-records"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $SHORT > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $SHORT > $OUT/pmc_write.json 2> $OUT/pmc_write.err
cd $R
python3 tools/pmc_summary.py $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1) $OUT/pmc_traffic.json > $OUT/pmc_summary.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/kt
# the float32 pass alone (warm-up + one step of 16 frames = 32 frames): traffic of the whole pass per frame
cd /tmp
SHORT32="--frames 16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --check-frames 0 --no-split-record --no-b3-record --no-config-records"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch32 -- python3 $R/bench.py $SHORT32 > $OUT/pmc_fetch32.json 2> $OUT/pmc_fetch32.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write32 -- python3 $R/bench.py $SHORT32 > $OUT/pmc_write32.json 2> $OUT/pmc_write32.err
cd $R
python3 tools/pmc_summary.py $(ls $OUT/pmc_fetch32/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_write32/*/*counter_collection.csv | head -1) $OUT/pmc_traffic_f32.json 32 > $OUT/pmc_summary_f32.txt
rm -rf $OUT/pmc_fetch32 $OUT/pmc_write32
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/pmc_summary.txt
timeout -k 10 600 python3 bench.py --two-streams --no-split-record --no-cpu-baseline > $OUT/bench_two_streams.json 2> $OUT/bench2.err
timeout -k 10 600 python3 bench.py --direct-record --no-split-record --no-cpu-baseline > $OUT/bench_direct.json 2> $OUT/bench3.err
