#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_bucket_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -5 || exit 1
cd tests
for seed in 11 12 13; do
  timeout -k 10 700 python -u fuzz_parity.py 150 $seed > $R/gpurun_out/r02_fuzz_$seed.log 2>&1; echo "seed $seed rc $? $(tail -1 $R/gpurun_out/r02_fuzz_$seed.log)"
done
