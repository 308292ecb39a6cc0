#!/bin/bash
# tools/pmc_profile.sh TAG -- HBM traffic per kernel: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate
# passes (counters only, with --kernel-trace) over one in-order step at the full launch size (1/10 of the reads),
# summed per kernel by tools/pmc_summary.py into gpurun_out/pmcTAG_summary.txt
set -o pipefail
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
export KBBQ_NO_OVERLAP=1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc${tag}_$c -- \
        python3 $R/bench.py --genome-len 300000000 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc${tag}_$c.log 2>&1 || exit 1
    echo "$c done"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc${tag}_FETCH_SIZE $R/gpurun_out/pmc${tag}_WRITE_SIZE > $R/gpurun_out/pmc${tag}_summary.txt
