#!/bin/bash
# bucketed inserts: parity first, then A/B at 1/10 scale (in order: exclusive kernel times), then full scale
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-a}
mkdir -p $R/gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_bucket_gpu.py -x -q 2>&1 | tee $R/gpurun_out/r02_bucket_tests_$tag.log | tail -15 || exit 1
echo "bucket tests done"
KBBQ_NO_OVERLAP=1 KBBQ_BUCKET=0 timeout -k 10 200 python bench.py --genome-len 300000000 --no-cpu-baseline --steps 2 > $R/gpurun_out/r02_ab_direct_$tag.json 2> $R/gpurun_out/r02_ab_$tag.log || exit 1
KBBQ_NO_OVERLAP=1 timeout -k 10 200 python bench.py --genome-len 300000000 --no-cpu-baseline --steps 2 > $R/gpurun_out/r02_ab_bucket_$tag.json 2>> $R/gpurun_out/r02_ab_$tag.log || exit 1
echo "ab done"
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_bench_full_bucket_$tag.json 2> $R/gpurun_out/r02_bench_full_bucket_$tag.log || exit 1
echo "full done"
KBBQ_NO_OVERLAP=1 timeout -k 10 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_bench_full_bucket_${tag}_inorder.json 2>> $R/gpurun_out/r02_bench_full_bucket_$tag.log || exit 1
echo "in-order done"
