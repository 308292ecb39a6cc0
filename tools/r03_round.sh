#!/bin/bash
# tools/r03_round.sh TAG -- the whole GPU suite, then bench.py with its default arguments (CPU baseline + PCIe-inclusive leg,
# asynchronous host batches) and the one-rank exchange at full size
set -o pipefail
tag=${1:-a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q > $R/gpurun_out/r03_gpu_tests_$tag.log 2>&1; rc=$?
tail -15 $R/gpurun_out/r03_gpu_tests_$tag.log
[ $rc -eq 0 ] || exit 1
echo "gpu tests done"
timeout -k 10 600 python $R/bench.py > $R/gpurun_out/r03_bench_full_$tag.json 2> $R/gpurun_out/r03_bench_full_$tag.log || { tail -5 $R/gpurun_out/r03_bench_full_$tag.log; exit 1; }
python3 - <<PY
import json
d=json.loads(open("$R/gpurun_out/r03_bench_full_$tag.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "k_infer", d["kernels"]["k_infer"]["avg_ms"])
p=d["pcie_inclusive"]
for k in ("resubmit","resubmit_synchronous_calls","upload_once"):
    print(k, p[k]["value"], p[k]["pass_seconds"], p[k]["fraction_of_bound"], p[k]["fraction_of_duplex_bound"])
print("cpu", d["cpu_baseline"]["value"])
PY
timeout -k 10 400 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie --force-exchange > $R/gpurun_out/r03_exchange_one_rank_$tag.json 2> $R/gpurun_out/r03_exchange_$tag.log || { tail -5 $R/gpurun_out/r03_exchange_$tag.log; exit 1; }
python3 -c "
import json
d=json.loads(open('$R/gpurun_out/r03_exchange_one_rank_$tag.json').read().strip().splitlines()[-1])
print(json.dumps(d['exchange_one_rank'], indent=1)); print('ms_per_step', d['ms_per_step'])"
