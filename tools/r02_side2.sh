#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-i}
mkdir -p $R/gpurun_out
for v in side noside side noside; do
  if [ $v = noside ]; then export KBBQ_NO_SIDE2=1; else unset KBBQ_NO_SIDE2; fi
  timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/r02_bench_full_${tag}_$v.json 2> $R/gpurun_out/r02_bench_full_$tag.log || exit 1
  python - <<PY
import json
d=json.loads(open("$R/gpurun_out/r02_bench_full_${tag}_$v.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["value"], d["bucketed_inserts"]["flushes_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items() if k in ("k_infer","k_emit_trusted","k_apply_trusted","k_split_trusted")})
PY
done
