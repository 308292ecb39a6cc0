#!/bin/bash
# tools/round_profile.sh TAG [ROUND] -- the round's profiler record for HEAD at full scale (BASELINE configs[1]); ROUND = file prefix (default r04):
#   gpurun_out/${rnd}_bench_driver_TAG.json        python bench.py --gpus 1 --steps 20 --warmup 5 (what the driver runs)
#   gpurun_out/${rnd}_kernel_stats_TAG.csv         rocprofv3 --kernel-trace --stats of one step of the same command
#   gpurun_out/${rnd}_kernel_stats_inorder_TAG.csv the same with KBBQ_NO_OVERLAP=1 (exclusive durations)
#   gpurun_out/${rnd}_pmc_TAG_summary.txt          rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of one in-order 3e8-genome step
#   gpurun_out/${rnd}_pmc_latest_TAG.json          the same as the JSON bench.py quotes (copy to profiles/${rnd}_pmc_latest.json)
set -o pipefail
tag=$1
rnd=${2:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof$tag -o ${rnd} -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-exclusive-step > $R/gpurun_out/prof$tag.json 2> $R/gpurun_out/prof$tag.log || { tail -5 $R/gpurun_out/prof$tag.log; exit 1; }
find $R/gpurun_out/prof$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${rnd}_kernel_stats_$tag.csv \;
rm -rf $R/gpurun_out/prof$tag
echo "profile done"
# the same with every kernel in order on one stream: the exclusive durations the roofline object quotes
KBBQ_NO_OVERLAP=1 timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profi$tag -o ${rnd} -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-exclusive-step > $R/gpurun_out/profi$tag.json 2> $R/gpurun_out/profi$tag.log || { tail -5 $R/gpurun_out/profi$tag.log; exit 1; }
find $R/gpurun_out/profi$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${rnd}_kernel_stats_inorder_$tag.csv \;
rm -rf $R/gpurun_out/profi$tag
echo "in-order profile done"
export KBBQ_NO_OVERLAP=1
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc${tag}_$c -- \
        python3 $R/bench.py --genome-len 300000000 --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-exclusive-step > $R/gpurun_out/pmc${tag}_$c.log 2>&1 || { tail -5 $R/gpurun_out/pmc${tag}_$c.log; exit 1; }
    echo "$c done"
done
unset KBBQ_NO_OVERLAP
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc${tag}_FETCH_SIZE $R/gpurun_out/pmc${tag}_WRITE_SIZE > $R/gpurun_out/${rnd}_pmc_${tag}_summary.txt
rm -rf $R/gpurun_out/pmc${tag}_FETCH_SIZE $R/gpurun_out/pmc${tag}_WRITE_SIZE
(cd $R && python3 tools/pmc_json.py gpurun_out/${rnd}_pmc_${tag}_summary.txt gpurun_out/${rnd}_pmc_latest_$tag.json profiles/${rnd}_pmc_fetch_write_$tag.txt)
echo "pmc done"
cd $R
timeout -k 10 600 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/${rnd}_bench_driver_$tag.json 2> $R/gpurun_out/${rnd}_bench_driver_$tag.log || { tail -5 $R/gpurun_out/${rnd}_bench_driver_$tag.log; exit 1; }
python3 -c "
import json
d=json.loads(open('$R/gpurun_out/${rnd}_bench_driver_$tag.json').read().strip().splitlines()[-1])
print('driver-like', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['traffic'])"
