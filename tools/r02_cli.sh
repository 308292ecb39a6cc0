#!/bin/bash
# CLI tests on the GPU, then the end-to-end runs: tools/r02_e2e.sh (FASTQ split) and/or tools/r02_e2e_bam.sh (BAM split + digests)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_cli_gpu.py -x -q 2>&1 | tee $R/gpurun_out/r02_cli_tests.log | tail -5 || exit 1
case "$1" in
  fastq) bash tools/r02_e2e.sh ${2:-100000000} ;;
  bam) bash tools/r02_e2e_bam.sh ${2:-100000000} ;;
esac
