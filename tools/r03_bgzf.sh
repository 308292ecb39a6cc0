#!/bin/bash
# tools/r03_bgzf.sh TAG -- the device BGZF writer's tests, then the step again (dynamic read chunks in k_infer / k_scan_trusted)
set -o pipefail
tag=${1:-a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_bgzf_gpu.py -x -q -s > $R/gpurun_out/r03_bgzf_tests_$tag.log 2>&1; rc=$?
tail -25 $R/gpurun_out/r03_bgzf_tests_$tag.log
echo "bgzf tests rc $rc"
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py tests/test_golden.py -m gpu -x -q > $R/gpurun_out/r03_parity_$tag.log 2>&1; rc2=$?
tail -5 $R/gpurun_out/r03_parity_$tag.log
[ $rc2 -eq 0 ] || exit 1
KBBQ_NO_OVERLAP=1 timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/r03_bench_${tag}_inorder.json 2> $R/gpurun_out/r03_bench_$tag.log || exit 1
timeout -k 10 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/r03_bench_${tag}.json 2>> $R/gpurun_out/r03_bench_$tag.log || exit 1
python3 $R/tools/ab_show.py $R/gpurun_out/r03_bench_${tag}_inorder.json $R/gpurun_out/r03_bench_${tag}.json
