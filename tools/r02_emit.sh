#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-j}
mkdir -p $R/gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_bucket_gpu.py tests/test_parity_gpu.py -x -q 2>&1 | tee $R/gpurun_out/r02_bucket_tests_$tag.log | tail -5 || exit 1
KBBQ_NO_OVERLAP=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/r02_bench_full_${tag}_inorder.json 2> $R/gpurun_out/r02_bench_full_$tag.log || exit 1
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/r02_bench_full_${tag}.json 2>> $R/gpurun_out/r02_bench_full_$tag.log || exit 1
python - <<PY
import json
for f in ("$R/gpurun_out/r02_bench_full_${tag}_inorder.json","$R/gpurun_out/r02_bench_full_${tag}.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(d["ms_per_step"], d["value"], d["bucketed_inserts"]["flushes_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
