#!/bin/bash
# round 2, first GPU call: parity suite, then the per-rank compute of an 8-rank job emulated on one GPU (in order and
# overlapped), then the full single-GPU line for the same box
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q 2>&1 | tee $R/gpurun_out/r02_gpu_tests_first.log | tail -5 || exit 1
for r in 0 3; do
KBBQ_NO_OVERLAP=1 timeout -k 10 300 python bench.py --emulate-shard $r/8 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_shard8_rank${r}_inorder.json 2> $R/gpurun_out/r02_shard8_rank${r}.log || exit 1
done
timeout -k 10 300 python bench.py --emulate-shard 0/8 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_shard8_rank0.json 2>> $R/gpurun_out/r02_shard8_rank0.log || exit 1
KBBQ_NO_OVERLAP=1 timeout -k 10 300 python bench.py --emulate-shard 0/2 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_shard2_rank0_inorder.json 2>> $R/gpurun_out/r02_shard8_rank0.log || exit 1
echo "shards done"
timeout -k 10 420 python bench.py --steps 2 --warmup 1 > $R/gpurun_out/r02_bench_full_first.json 2> $R/gpurun_out/r02_bench_full_first.log || exit 1
echo "full done"
KBBQ_NO_OVERLAP=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_bench_full_first_inorder.json 2>> $R/gpurun_out/r02_bench_full_first.log || exit 1
echo "in-order done"
