#!/bin/bash
# pass 2: only the emits on the side stream (KBBQ_PASS2_SIDE=2) against everything in order (default) and the whole insert side there (=1)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-a}
mkdir -p $R/gpurun_out
for v in ${VARIANTS:-2 0 2 0 1}; do
  export KBBQ_PASS2_SIDE=$v
  timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/r02_side3_${tag}_$v.json 2> $R/gpurun_out/r02_side3_$tag.log || { tail -5 $R/gpurun_out/r02_side3_$tag.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("$R/gpurun_out/r02_side3_${tag}_$v.json").read().strip().splitlines()[-1])
print("side=$v", d["ms_per_step"], d["value"], d["result"]["recal_qual_sum"], d["bucketed_inserts"]["flushes_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if k in ("k_infer","k_emit_trusted","k_apply_trusted","k_split_trusted")})
PY
done
