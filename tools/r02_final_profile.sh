#!/bin/bash
# tools/r02_final_profile.sh TAG -- the round's record for HEAD at full scale (BASELINE configs[1]):
#   gpurun_out/r02_bench_full_TAG.json          python bench.py (default arguments: CPU baseline + PCIe-inclusive leg)
#   gpurun_out/r02_bench_full_TAG_inorder.json  the same build with KBBQ_NO_OVERLAP=1 (exclusive kernel durations)
#   gpurun_out/r02_kernel_stats_TAG.csv         rocprofv3 --kernel-trace --stats of one step of the same command
#   gpurun_out/r02_pmc_TAG_summary.txt          rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of one in-order 1/10-scale step
set -o pipefail
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 600 python $R/bench.py > $R/gpurun_out/r02_bench_full_$tag.json 2> $R/gpurun_out/r02_bench_full_$tag.log || exit 1
echo "bench done"
KBBQ_NO_OVERLAP=1 timeout -k 10 300 python $R/bench.py --no-cpu-baseline --no-pcie > $R/gpurun_out/r02_bench_full_${tag}_inorder.json 2>> $R/gpurun_out/r02_bench_full_$tag.log || exit 1
echo "in-order bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof$tag -o r02 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie > $R/gpurun_out/prof$tag.json 2> $R/gpurun_out/prof$tag.log || exit 1
find $R/gpurun_out/prof$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r02_kernel_stats_$tag.csv \;
rm -rf $R/gpurun_out/prof$tag
echo "profile done"
export KBBQ_NO_OVERLAP=1
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc${tag}_$c -- \
        python3 $R/bench.py --genome-len 300000000 --steps 1 --warmup 0 --no-cpu-baseline --no-pcie > $R/gpurun_out/pmc${tag}_$c.log 2>&1 || exit 1
    echo "$c done"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc${tag}_FETCH_SIZE $R/gpurun_out/pmc${tag}_WRITE_SIZE > $R/gpurun_out/r02_pmc_${tag}_summary.txt
rm -rf $R/gpurun_out/pmc${tag}_FETCH_SIZE $R/gpurun_out/pmc${tag}_WRITE_SIZE
echo "pmc done"
