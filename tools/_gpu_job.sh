set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout -k 10 900 python -u -m pytest tests/test_gpu_production.py tests/test_gpu_configs.py::test_4k_frame_matches_oracle tests/test_gpu_network.py -x -v --timeout 600 --timeout-method thread > gpurun_out/r2e/prod.log 2>&1 || { tail -60 gpurun_out/r2e/prod.log; exit 1; }
grep -E "PASSED|FAILED|passed|failed" gpurun_out/r2e/prod.log | tail -40
