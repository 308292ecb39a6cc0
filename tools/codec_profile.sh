#!/bin/bash
# tools/r03_codec_profile.sh -- rocprofv3 --kernel-trace --stats of the two codec probes (k_deflate, k_inflate, k_block_crc and the
# index kernels): gpurun_out/r03_kernel_stats_codec_{deflate,inflate}.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in deflate inflate; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profc_$w -o r03 -- python3 $R/tools/${w}_probe.py 64 8 > $R/gpurun_out/profc_$w.log 2>&1 || { tail -5 $R/gpurun_out/profc_$w.log; exit 1; }
    find $R/gpurun_out/profc_$w -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r03_kernel_stats_codec_$w.csv \;
    rm -rf $R/gpurun_out/profc_$w
    head -6 $R/gpurun_out/r03_kernel_stats_codec_$w.csv | cut -c1-160
done
