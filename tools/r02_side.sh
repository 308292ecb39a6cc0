#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-h}
mkdir -p $R/gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q 2>&1 | tee $R/gpurun_out/r02_gpu_tests_$tag.log | tail -15 || exit 1
echo "gpu tests done"
timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/r02_bench_full_$tag.json 2> $R/gpurun_out/r02_bench_full_$tag.log || exit 1
echo "full done"
KBBQ_NO_OVERLAP=1 timeout -k 10 500 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/r02_bench_full_${tag}_inorder.json 2>> $R/gpurun_out/r02_bench_full_$tag.log || exit 1
echo "in-order done"
