set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout -k 10 600 python -u -m pytest tests/test_gpu_precision.py -x -v -s --timeout 300 --timeout-method thread > gpurun_out/r2c/prec.log 2>&1 || { tail -40 gpurun_out/r2c/prec.log; exit 1; }
tail -25 gpurun_out/r2c/prec.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --deselect tests/test_gpu_precision.py > gpurun_out/r2c/all.log 2>&1 || { tail -40 gpurun_out/r2c/all.log; exit 1; }
tail -5 gpurun_out/r2c/all.log
