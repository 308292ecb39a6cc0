#!/bin/bash
# occupancy variants of the two lookup kernels, 1/10 scale, in order (exclusive kernel durations), A/B/A/B in one job
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for rep in 1 2; do
for v in "0 0" "8 8" "7 8" "0 8" "8 0"; do
  set -- $v
  KBBQ_NO_OVERLAP=1 KBBQ_INFER_OCC=$1 KBBQ_SCAN_OCC=$2 timeout -k 10 200 python bench.py --genome-len 300000000 --no-cpu-baseline --no-pcie --steps 2 > $R/gpurun_out/r02_occ_$1_$2.json 2>> $R/gpurun_out/r02_occ.log || exit 1
  python - <<PY
import json
d=json.loads(open("$R/gpurun_out/r02_occ_$1_$2.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("infer_occ $1 scan_occ $2: step", d["ms_per_step"], "k_infer", k["k_infer"]["avg_ms"], "k_scan", k["k_scan_trusted"]["avg_ms"], "digest", d["result"]["recal_qual_sum"])
PY
done
done
