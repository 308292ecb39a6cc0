#!/bin/bash
# tools/r03_first.sh TAG -- GPU parity suite, then the 30x step in order (exclusive kernel durations) and overlapped
set -o pipefail
tag=${1:-a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q > $R/gpurun_out/r03_gpu_tests_$tag.log 2>&1; rc=$?
tail -15 $R/gpurun_out/r03_gpu_tests_$tag.log
[ $rc -eq 0 ] || exit 1
echo "gpu tests done"
KBBQ_NO_OVERLAP=1 timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/r03_bench_${tag}_inorder.json 2> $R/gpurun_out/r03_bench_$tag.log || exit 1
echo "in-order bench done"
timeout -k 10 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/r03_bench_${tag}.json 2>> $R/gpurun_out/r03_bench_$tag.log || exit 1
python3 $R/tools/ab_show.py $R/gpurun_out/r03_bench_${tag}_inorder.json $R/gpurun_out/r03_bench_${tag}.json 2>/dev/null || tail -c 1500 $R/gpurun_out/r03_bench_${tag}.json
