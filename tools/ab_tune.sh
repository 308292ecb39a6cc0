#!/bin/bash
# tools/ab_tune.sh KNOB TAG [ROUNDS] [TESTS] -- optional parity tests (pytest -k TESTS), then KNOB=0/1 alternated inside one
# bench.py process at full scale (bench.py --ab-tune): gpurun_out/ab_tune_TAG.json
set -o pipefail
knob=$1; tag=$2; rounds=${3:-4}; tests=$4
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
if [ -n "$tests" ]; then
    timeout -k 10 600 python -u -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "$tests" > $R/gpurun_out/ab_tune_${tag}_tests.log 2>&1 || { tail -30 $R/gpurun_out/ab_tune_${tag}_tests.log; exit 1; }
    tail -2 $R/gpurun_out/ab_tune_${tag}_tests.log
fi
timeout -k 10 600 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie --ab-tune $knob,$rounds > $R/gpurun_out/ab_tune_$tag.json 2> $R/gpurun_out/ab_tune_$tag.log || { tail -5 $R/gpurun_out/ab_tune_$tag.log; exit 1; }
grep "\[ab\]" $R/gpurun_out/ab_tune_$tag.log
