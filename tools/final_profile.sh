#!/bin/bash
# tools/final_profile.sh TAG -- the round's record for HEAD at full scale (BASELINE configs[1]):
#   gpurun_out/bench_full_TAG.json          python bench.py (default arguments, with the CPU baseline)
#   gpurun_out/bench_full_TAG_inorder.json  the same build with KBBQ_NO_OVERLAP=1 (exclusive kernel durations)
#   gpurun_out/profTAG/                     rocprofv3 --kernel-trace --stats of one step of the same command
set -o pipefail
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 420 python $R/bench.py > $R/gpurun_out/bench_full_$tag.json 2> $R/gpurun_out/bench_full_$tag.log || exit 1
echo "bench done"
KBBQ_NO_OVERLAP=1 timeout -k 10 300 python $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_full_${tag}_inorder.json 2>> $R/gpurun_out/bench_full_$tag.log || exit 1
echo "in-order bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof$tag -o r01 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/prof$tag.json 2> $R/gpurun_out/prof$tag.log || exit 1
echo "profile done"
