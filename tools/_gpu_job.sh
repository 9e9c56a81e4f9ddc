set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout -k 10 900 python -u -m pytest tests/test_gpu_network.py tests/test_gpu_production.py tests/test_gpu_precision.py tests/test_gpu_units.py -x -q --timeout 600 --timeout-method thread > gpurun_out/r2i/t.log 2>&1 || { tail -60 gpurun_out/r2i/t.log; exit 1; }
tail -3 gpurun_out/r2i/t.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-split-record > gpurun_out/r2i/f32_new.json 2> gpurun_out/r2i/f32_new.err
FD_F32_LAYERWISE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-split-record > gpurun_out/r2i/f32_old.json 2> gpurun_out/r2i/f32_old.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-split-record --streams 2 > gpurun_out/r2i/f32_new2s.json 2> gpurun_out/r2i/f32_new2s.err
python - <<'PY'
import json
for f in ("f32_new","f32_old","f32_new2s"):
    d=json.load(open("gpurun_out/r2i/%s.json"%f))
    r=d["roofline"]
    print(f, d["value"], r["kernel"], r["achieved"], r["frac"], r["pipeline_frac"])
    for k in d["kernels_single_stream_step"][:7]: print("   ",k)
PY
