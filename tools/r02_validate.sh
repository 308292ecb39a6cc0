#!/bin/bash
# what the driver runs at the end of the round, in one call: the GPU tests, smoke(), the default bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-v}
mkdir -p $R/gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q 2>&1 | tee $R/gpurun_out/r02_gpu_tests_$tag.log | tail -6 || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 || exit 1
timeout -k 10 900 python bench.py --gpus 1 --steps 5 --warmup 2 > $R/gpurun_out/r02_bench_full_$tag.json 2> $R/gpurun_out/r02_bench_full_$tag.log || exit 1
python - <<PY
import json
d=json.loads(open("$R/gpurun_out/r02_bench_full_$tag.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["bucketed_inserts"], d["cpu_baseline"]["value"])
print(json.dumps(d["pcie_inclusive"]["resubmit"]), json.dumps(d["pcie_inclusive"]["upload_once"]))
print({k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
