#!/bin/bash
# tools/sq_deflate.sh TAG -- SQ counters of k_deflate (tools/deflate_probe.py): which issue port the encoder fills.
# gpurun_out/sqdfl<TAG>_summary.txt
set -o pipefail
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/sqdfl${tag}_$i -- \
        python3 $R/tools/deflate_probe.py 64 8 > $R/gpurun_out/sqdfl${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/sqdfl${tag}_$i.log; }
    echo "pass $i done"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/sqdfl${tag}_1 $R/gpurun_out/sqdfl${tag}_2 $R/gpurun_out/sqdfl${tag}_3 $R/gpurun_out/sqdfl${tag}_4 > $R/gpurun_out/sqdfl${tag}_summary.txt
grep -h "GB/s" $R/gpurun_out/sqdfl${tag}_1.log >> $R/gpurun_out/sqdfl${tag}_summary.txt
rm -rf $R/gpurun_out/sqdfl${tag}_1 $R/gpurun_out/sqdfl${tag}_2 $R/gpurun_out/sqdfl${tag}_3 $R/gpurun_out/sqdfl${tag}_4
