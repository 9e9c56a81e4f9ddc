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
cp $OUT/pmc_traffic_f32.json $R/profiles/${TAG}_pmc_traffic_f32.json
bash tools/profile_round.sh $TAG
timeout -k 10 200 python3 tools/frontend_bench.py 64 > $OUT/frontend.json 2>/dev/null
timeout -k 10 200 python3 tools/frontend_bench.py 128 > $OUT/frontend_128.json 2>/dev/null
timeout -k 10 300 python3 tools/time_configs.py f16x3 > $OUT/cfg_split.txt 2>&1
timeout -k 10 300 python3 tools/time_configs.py f32 > $OUT/cfg_f32.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt2 -- python3 $R/bench.py --frames 16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --check-frames 0 --no-split-record --no-b3-record --no-config-records > $OUT/b16.json 2> $OUT/b16.err
cd $R
python3 tools/conv_layer_report.py $(ls $OUT/kt2/*/*kernel_trace.csv | head -1) 16 f32 > $OUT/layers_f32.txt 2>&1 || true
python3 tools/ridge_table.py $OUT/layers_f32.txt 16 > $OUT/ridge_f32.txt 2>&1 || true
rm -rf $OUT/kt2
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt3 -- python3 $R/bench.py --precision f16x3 --streams 1 --frames 16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --check-frames 0 --no-split-record --no-b3-record --no-config-records > $OUT/b16s.json 2> $OUT/b16s.err
cd $R
python3 tools/conv_layer_report.py $(ls $OUT/kt3/*/*kernel_trace.csv | head -1) 16 f16x3 > $OUT/layers_f16x3.txt 2>&1 || true
rm -rf $OUT/kt3
# SQ counters of one float32 micro-batch (one --pmc pass), HBM traffic of the front-end kernels (two passes), and the
# float32 pass with the frames uploaded from pinned host memory inside the timed region
cd /tmp
SHORT="--frames 16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --check-frames 0 --no-split-record --no-b3-record --no-config-records"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/bench.py $SHORT > $OUT/sq.json 2> $OUT/sq.err
cp $(ls $OUT/sq/*/*counter_collection.csv | head -1) $OUT/sq_counters_f32.csv
rm -rf $OUT/sq
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rz_fetch -- python3 $R/tools/frontend_bench.py 64 > $OUT/rz_fetch.json 2> $OUT/rz_fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/rz_write -- python3 $R/tools/frontend_bench.py 64 > $OUT/rz_write.json 2> $OUT/rz_write.err
cd $R
python3 tools/pmc_summary.py $(ls $OUT/rz_fetch/*/*counter_collection.csv | head -1) $(ls $OUT/rz_write/*/*counter_collection.csv | head -1) $OUT/frontend_pmc_traffic.json > $OUT/rz_pmc_summary.txt
rm -rf $OUT/rz_fetch $OUT/rz_write
timeout -k 10 400 python3 bench.py --from-host --no-split-record --no-cpu-baseline --no-profile > $OUT/bench_from_host.json 2> $OUT/bench_from_host.err
python3 tools/sq_summary.py $OUT/sq_counters_f32.csv 14 > $OUT/sq_table.md
echo final round done
