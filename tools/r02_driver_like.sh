#!/bin/bash
# the driver's N = 1 command, timed: python3 bench.py --gpus 1 --steps 20 --warmup 5
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
S=$(date +%s)
timeout -k 10 800 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r02_bench_driver_like.json 2> $R/gpurun_out/r02_bench_driver_like.log
rc=$?
E=$(date +%s)
echo "rc $rc wall $((E-S)) s"
python3 - <<PY
import json
d=json.loads(open("$R/gpurun_out/r02_bench_driver_like.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["steps"], d["warmup"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["n_gpus"], d["scaling"])
PY
