set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/r2g/all.log 2>&1 || { tail -60 gpurun_out/r2g/all.log; exit 1; }
tail -5 gpurun_out/r2g/all.log
