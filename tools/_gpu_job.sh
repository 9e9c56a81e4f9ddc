set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout -k 10 600 python -u -m pytest tests/test_gpu_threads.py tests/test_gpu_pipeline.py tests/test_gpu_boxes.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2f/t.log 2>&1 || { tail -60 gpurun_out/r2f/t.log; exit 1; }
tail -5 gpurun_out/r2f/t.log
