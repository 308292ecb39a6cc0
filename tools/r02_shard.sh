#!/bin/bash
# per-rank compute of the N-rank strong-scaling job emulated on one GPU (bench.py --emulate-shard), bucketed and direct inserts
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-g}
mkdir -p $R/gpurun_out
for n in 8 4 2; do
timeout -k 10 400 python bench.py --emulate-shard 0/$n --steps 1 --warmup 1 > $R/gpurun_out/r02_shard${n}_rank0_$tag.json 2> $R/gpurun_out/r02_shard_$tag.log || exit 1
echo "shard 0/$n done"
done
timeout -k 10 400 python bench.py --emulate-shard 5/8 --steps 1 --warmup 1 > $R/gpurun_out/r02_shard8_rank5_$tag.json 2>> $R/gpurun_out/r02_shard_$tag.log || exit 1
KBBQ_BUCKET=0 timeout -k 10 400 python bench.py --emulate-shard 0/8 --steps 1 --warmup 1 > $R/gpurun_out/r02_shard8_rank0_direct_$tag.json 2>> $R/gpurun_out/r02_shard_$tag.log || exit 1
echo "all done"
