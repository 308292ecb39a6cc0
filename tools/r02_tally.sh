#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q 2>&1 | tee $R/gpurun_out/r02_gpu_tests_tally.log | tail -6 || exit 1
for rep in 1 2; do
for v in general uniform; do
  if [ $v = general ]; then export KBBQ_TALLY_GENERAL=1; else unset KBBQ_TALLY_GENERAL; fi
  KBBQ_NO_OVERLAP=1 timeout -k 10 200 python bench.py --genome-len 300000000 --no-cpu-baseline --no-pcie --steps 2 > $R/gpurun_out/r02_tally_$v.json 2>> $R/gpurun_out/r02_tally.log || exit 1
  python - <<PY
import json
d=json.loads(open("$R/gpurun_out/r02_tally_$v.json").read().strip().splitlines()[-1])
print("$v: step", d["ms_per_step"], "k_tally", d["kernels"]["k_tally"]["avg_ms"], "digest", d["result"]["recal_qual_sum"])
PY
done
done
