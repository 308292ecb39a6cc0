#!/bin/bash
# emit variants at 1/10 scale, in order (exclusive kernel times): private regions (default), RPW=8, XCD-shared regions
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-c}
mkdir -p $R/gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_bucket_gpu.py -x -q 2>&1 | tee $R/gpurun_out/r02_bucket_tests_$tag.log | tail -5 || exit 1
KBBQ_BUCKET_SHARED=1 timeout -k 10 600 python -u -m pytest tests/test_bucket_gpu.py -x -q 2>&1 | tee -a $R/gpurun_out/r02_bucket_tests_$tag.log | tail -5 || exit 1
echo "bucket tests done"
KBBQ_NO_OVERLAP=1 timeout -k 10 200 python bench.py --genome-len 300000000 --no-cpu-baseline --steps 2 > $R/gpurun_out/r02_ab_priv_$tag.json 2> $R/gpurun_out/r02_ab_$tag.log || exit 1
KBBQ_NO_OVERLAP=1 KBBQ_EMIT_RPW=8 timeout -k 10 200 python bench.py --genome-len 300000000 --no-cpu-baseline --steps 2 > $R/gpurun_out/r02_ab_priv8_$tag.json 2>> $R/gpurun_out/r02_ab_$tag.log || exit 1
KBBQ_NO_OVERLAP=1 KBBQ_BUCKET_SHARED=1 timeout -k 10 200 python bench.py --genome-len 300000000 --no-cpu-baseline --steps 2 > $R/gpurun_out/r02_ab_shared_$tag.json 2>> $R/gpurun_out/r02_ab_$tag.log || exit 1
echo "ab done"
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_bench_full_bucket_$tag.json 2> $R/gpurun_out/r02_bench_full_bucket_$tag.log || exit 1
echo "full done"
