#!/bin/bash
# CLI tests, then the end-to-end split (tools/r02_e2e.sh GENOME_LEN: 30x of it, default 1e8 -> 3e9 bases)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
if [ "$1" != "e2e" ]; then
timeout -k 10 900 python -u -m pytest tests/test_cli_gpu.py -x -q 2>&1 | tee $R/gpurun_out/r02_cli_tests.log | tail -15 || exit 1
fi
bash tools/r02_e2e.sh ${2:-100000000}
