#!/bin/bash
# tools/r03_stats_only.sh TAG -- the kernel-trace part of tools/r03_final_profile.sh alone (overlapped and in-order step) and the driver's command
set -o pipefail
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in inorder overlapped; do
    if [ $mode = inorder ]; then export KBBQ_NO_OVERLAP=1; else unset KBBQ_NO_OVERLAP; fi
    timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profs_$mode -o r03 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-exclusive-step > $R/gpurun_out/profs_$mode.json 2> $R/gpurun_out/profs_$mode.log || { tail -5 $R/gpurun_out/profs_$mode.log; exit 1; }
    find $R/gpurun_out/profs_$mode -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r03_kernel_stats_${mode}_$tag.csv \;
    rm -rf $R/gpurun_out/profs_$mode
done
unset KBBQ_NO_OVERLAP
cd $R
timeout -k 10 600 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/r03_bench_driver_$tag.json 2> $R/gpurun_out/r03_bench_driver_$tag.log || { tail -5 $R/gpurun_out/r03_bench_driver_$tag.log; exit 1; }
python3 -c "
import json,csv
d=json.loads(open('$R/gpurun_out/r03_bench_driver_$tag.json').read().strip().splitlines()[-1]); r=d['roofline']
print('driver-like', d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['avg_launch_ms_before_timed_region'], r['traffic'])
for m in ('inorder','overlapped'):
    for row in csv.DictReader(open('$R/gpurun_out/r03_kernel_stats_%s_$tag.csv' % m)):
        if 'k_infer' in row['Name'].split('(')[0]: print(m, 'k_infer', round(float(row['AverageNs'])/1e6,3))"
