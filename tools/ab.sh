#!/bin/bash
# tools/ab.sh TAG -- GPU parity suite, then the 1/10-scale bench of the library in the tree; results in gpurun_out/ab_TAG.json
set -o pipefail
tag=$1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/gpu_tests_$tag.log | tail -4 || exit 1
timeout -k 10 200 python bench.py --genome-len 300000000 --no-cpu-baseline --steps 2 > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.log
echo bench rc $?
