#!/bin/bash
# A/B: the insert side of pass 2 beside k_infer (default) or behind it (KBBQ_PASS2_SIDE=0), everything else as shipped
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for v in 1 0 1 0; do
  KBBQ_PASS2_SIDE=$v timeout -k 10 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/ab_side$v.json 2>> $R/gpurun_out/ab_side.log || exit 1
  python3 -c "
import json
d=json.loads(open('$R/gpurun_out/ab_side$v.json').read().strip().splitlines()[-1])
print('side=$v', d['ms_per_step'], d['kernels']['k_infer']['avg_ms'], d['roofline']['frac'])"
done
