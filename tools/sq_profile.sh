#!/bin/bash
# tools/sq_profile.sh TAG -- SQ counters per kernel (one in-order 1/10-scale step): where the waves' cycles go
set -o pipefail
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
export KBBQ_NO_OVERLAP=1
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/sq${tag}_$i -- \
        python3 $R/bench.py --genome-len 300000000 --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-exclusive-step > $R/gpurun_out/sq${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/sq${tag}_$i.log; }
    echo "pass $i done"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/sq${tag}_1 $R/gpurun_out/sq${tag}_2 $R/gpurun_out/sq${tag}_3 $R/gpurun_out/sq${tag}_4 > $R/gpurun_out/sq${tag}_summary.txt
# keep the merge small: only the summaries travel back
rm -rf $R/gpurun_out/sq${tag}_1 $R/gpurun_out/sq${tag}_2 $R/gpurun_out/sq${tag}_3 $R/gpurun_out/sq${tag}_4
