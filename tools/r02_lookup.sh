#!/bin/bash
# slice-bucketed lookups (lookup.h): correctness of the chain at a small size, then its kernel times at full scale
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-a}
mkdir -p $R/gpurun_out
KBBQ_BENCH_GENOME=100000000 KBBQ_BENCH_BATCH=1048576 KBBQ_LOOKUP_PROBE=2 KBBQ_BUCKET=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie \
  > $R/gpurun_out/r02_lookup_small_$tag.json 2> $R/gpurun_out/r02_lookup_small_$tag.log || { tail -5 $R/gpurun_out/r02_lookup_small_$tag.log; exit 1; }
grep "lookup probe" $R/gpurun_out/r02_lookup_small_$tag.log | sort | uniq -c | sort -rn | head -8
KBBQ_LOOKUP_PROBE=1 timeout -k 10 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie \
  > $R/gpurun_out/r02_lookup_full_$tag.json 2> $R/gpurun_out/r02_lookup_full_$tag.log || { tail -5 $R/gpurun_out/r02_lookup_full_$tag.log; exit 1; }
grep "lookup probe" $R/gpurun_out/r02_lookup_full_$tag.log | head -3
python - <<PY
import json
d=json.loads(open("$R/gpurun_out/r02_lookup_full_$tag.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"])
for k,v in d["kernels"].items():
    if k.startswith("probe") or k in ("k_infer","k_emit_trusted"): print(k, v["launches"], v["avg_ms"])
PY
