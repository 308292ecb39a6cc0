#!/bin/bash
# A/B of the emit kernels: dense (marked k-mer windows gathered first, full wavefronts hashed) against hashing every lane
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-a}
mkdir -p $R/gpurun_out
for v in ${VARIANTS:-dense all dense all}; do
  unset KBBQ_EMIT_DENSE KBBQ_EMIT_RPW; if [ $v = all ]; then export KBBQ_EMIT_DENSE=0; fi; if [ $v = rpw8 ]; then export KBBQ_EMIT_RPW=8; fi; if [ $v = rpw2 ]; then export KBBQ_EMIT_RPW=2; fi
  timeout -k 10 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/r02_dense_${tag}_$v.json 2> $R/gpurun_out/r02_dense_$tag.log || { tail -5 $R/gpurun_out/r02_dense_$tag.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("$R/gpurun_out/r02_dense_${tag}_$v.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["value"], d["result"]["recal_qual_sum"], {k:v["avg_ms"] for k,v in d["kernels"].items() if k in ("k_emit_sampled","k_emit_trusted","k_draw_mask","k_infer")})
PY
done
