#!/bin/bash
# tools/cli_profile.sh TAG GENOME_LEN CASE -- rocprofv3 --kernel-trace --stats of ONE command-line run on the bench's own reads
# as a file (tools/e2e.sh's cases: fastq, bam, bam_setoq, bamoq_useoq_setoq), output to /dev/null:
# gpurun_out/cli_kernel_stats_TAG.csv, and the same run without the profiler for its phase split.
set -o pipefail
tag=$1; G=${2:-100000000}; c=${3:-bam_setoq}
D=${TMPDIR:-/tmp}/kbbq_cli_prof
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $D $R/gpurun_out
case $c in
    fastq) $R/kbbq_amd/kbbq --io-test synth-fastq $G 30 > $D/in; args="-g $G" ;;
    bam) $R/kbbq_amd/kbbq --io-test synth-bam $G 30 > $D/in; args="" ;;
    bam_setoq) $R/kbbq_amd/kbbq --io-test synth-bam $G 30 > $D/in; args="--set-oq" ;;
    bamoq_useoq_setoq) $R/kbbq_amd/kbbq --io-test synth-bam $G 30 oq > $D/in; args="--use-oq --set-oq" ;;
esac
ls -l $D/in
export KBBQ_TIMING=1 KBBQ_QUAL_DIGEST=1 KBBQ_SEED=777
for i in 1 2; do
    s=$(date +%s%N)
    $R/kbbq_amd/kbbq $args $D/in 2> $D/err.txt > /dev/null || { tail -3 $D/err.txt; exit 1; }
    e=$(date +%s%N)
    echo "run $i ($c $args, output to /dev/null) wall_ms $(( (e - s) / 1000000 ))"
    grep -E "timing|digest" $D/err.txt
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cliprof_$tag -o p -- $R/kbbq_amd/kbbq $args $D/in > /dev/null 2> $D/err_prof.txt || { tail -5 $D/err_prof.txt; exit 1; }
find $R/gpurun_out/cliprof_$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/cli_kernel_stats_$tag.csv \;
rm -rf $R/gpurun_out/cliprof_$tag $D
head -30 $R/gpurun_out/cli_kernel_stats_$tag.csv | cut -c1-150
