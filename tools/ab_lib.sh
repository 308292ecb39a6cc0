#!/bin/bash
# tools/ab_lib.sh VARIANT.so [TAG] -- the library in the tree against another build of it (KBBQ_LIB), in-order steps at
# full scale, base / variant / base / variant in one job: exclusive kernel durations and the step
set -o pipefail
V=$1; tag=${2:-ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in ${ROUNDS:-1 2}; do
  for which in base variant; do
    if [ $which = variant ]; then export KBBQ_LIB=$R/$V; else unset KBBQ_LIB; fi
    KBBQ_NO_OVERLAP=1 timeout -k 10 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-exclusive-step > $R/gpurun_out/ab_${tag}_${which}_$i.json 2> $R/gpurun_out/ab_${tag}_${which}_$i.log || { tail -3 $R/gpurun_out/ab_${tag}_${which}_$i.log; exit 1; }
    python3 -c "
import json
d=json.loads(open('$R/gpurun_out/ab_${tag}_${which}_$i.json').read().strip().splitlines()[-1])
k=d['kernels']
print('$which $i', 'step', d['ms_per_step'], 'k_infer', k['k_infer']['avg_ms'], 'k_scan', k['k_scan_trusted']['avg_ms'], 'walk', k['k_correct_wave']['avg_ms'], 'emit', k['k_emit_sampled']['avg_ms'], k['k_emit_trusted']['avg_ms'], 'split', k['k_split_trusted']['avg_ms'], 'apply', k['k_apply_trusted']['avg_ms'], 'digest', d['result']['recal_qual_sum'])"
  done
done
