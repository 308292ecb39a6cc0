#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for rep in 1 2; do
for v in "0 0" "6 6" "5 5" "4 4" "3 3" "8 8"; do
  set -- $v
  KBBQ_NO_OVERLAP=1 KBBQ_INFER_BLOCKS=$1 KBBQ_SCAN_BLOCKS=$2 timeout -k 10 200 python bench.py --genome-len 300000000 --no-cpu-baseline --no-pcie --steps 2 > $R/gpurun_out/r02_blk_$1_$2.json 2>> $R/gpurun_out/r02_occ.log || exit 1
  python - <<PY
import json
d=json.loads(open("$R/gpurun_out/r02_blk_$1_$2.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("infer_blocks/CU $1 scan_blocks/CU $2: step", d["ms_per_step"], "k_infer", k["k_infer"]["avg_ms"], "k_scan", k["k_scan_trusted"]["avg_ms"])
PY
done
done
