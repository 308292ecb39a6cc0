#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-f}
mkdir -p $R/gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q 2>&1 | tee $R/gpurun_out/r02_gpu_tests_$tag.log | tail -25 || exit 1
echo "gpu tests done"
