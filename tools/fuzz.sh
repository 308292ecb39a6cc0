#!/bin/bash
# tools/fuzz.sh -- randomised differential runs (tests/fuzz_parity.py) on the GPU; KBBQ_BUCKET=1 sends the small filters of these cases
# through the slice-bucketed insert path (emit / split / apply) as well
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd tests
for seed in ${SEEDS:-31 32}; do
  KBBQ_BUCKET=1 timeout -k 10 700 python -u fuzz_parity.py 150 $seed > $R/gpurun_out/fuzz_bucket_$seed.log 2>&1; echo "bucketed seed $seed rc $? $(tail -1 $R/gpurun_out/fuzz_bucket_$seed.log)"
done
for seed in ${SEEDS2:-33}; do
  timeout -k 10 700 python -u fuzz_parity.py 150 $seed > $R/gpurun_out/fuzz_$seed.log 2>&1; echo "direct seed $seed rc $? $(tail -1 $R/gpurun_out/fuzz_$seed.log)"
done
# the BAM path on the device against the host codec (tests/fuzz_bam.py)
for seed in ${SEEDS3:-51}; do
  timeout -k 10 900 python -u fuzz_bam.py ${BAM_CASES:-120} $seed > $R/gpurun_out/fuzz_bam_$seed.log 2>&1; echo "bam seed $seed rc $? $(tail -1 $R/gpurun_out/fuzz_bam_$seed.log)"
done
# k_inflate against zlib on arbitrary bytes and every kind of DEFLATE stream (tests/fuzz_inflate.py)
for seed in ${SEEDS4:-72}; do
  timeout -k 10 600 python -u fuzz_inflate.py ${INFLATE_CASES:-300} $seed > $R/gpurun_out/fuzz_inflate_$seed.log 2>&1; echo "inflate seed $seed rc $? $(tail -1 $R/gpurun_out/fuzz_inflate_$seed.log)"
done
