#!/bin/bash
# tools/ab_steps.sh TAG [TESTS] -- optional parity tests, then one in-order and one overlapped full-scale step of the library in
# the tree: exclusive kernel durations and the step (compare with the last committed profile)
set -o pipefail
tag=$1; tests=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ -n "$tests" ]; then
    timeout -k 10 900 python -u -m pytest tests -m gpu -x -q -k "$tests" > $R/gpurun_out/ab_steps_${tag}_tests.log 2>&1 || { tail -30 $R/gpurun_out/ab_steps_${tag}_tests.log; exit 1; }
    tail -2 $R/gpurun_out/ab_steps_${tag}_tests.log
fi
timeout -k 10 400 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > $R/gpurun_out/ab_steps_$tag.json 2> $R/gpurun_out/ab_steps_$tag.log || { tail -5 $R/gpurun_out/ab_steps_$tag.log; exit 1; }
python3 -c "
import json
d=json.loads(open('$R/gpurun_out/ab_steps_$tag.json').read().strip().splitlines()[-1])
print('step', d['ms_per_step'], 'value', d['value'], 'digest', d['result']['recal_qual_sum'])
print({k:(v.get('exclusive_avg_ms'), v['avg_ms']) for k,v in d['kernels'].items()})"
