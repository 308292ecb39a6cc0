#!/bin/bash
# tools/r03_e2e.sh [GENOME_LEN] [TAG] -- the command line end to end with the per-phase split (KBBQ_TIMING=1): bgzip-ed
# FASTQ in (block-parallel inflate), BGZF FASTQ out through the encoder on the GPU (default) and through host zlib
# (KBBQ_HOST_DEFLATE=1: rounds 1-2) and in streaming mode; outputs must decompress to the same bytes.  gpurun_out/r03_e2e_TAG.log
set -o pipefail
G=${1:-100000000}
tag=${2:-a}
D=${TMPDIR:-/tmp}/kbbq_e2e
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/gpurun_out/r03_e2e_$tag.log
mkdir -p $D $R/gpurun_out
: > $L
nproc >> $L
python tools/make_fastq.py $D/big.fq $G 30 >> $L 2>&1 || exit 1
$R/kbbq_amd/kbbq --io-test bgzf 16 < $D/big.fq > $D/big.fq.gz || exit 1
ls -l $D/big.fq $D/big.fq.gz >> $L
run() {   # name, env..., -- args
    local name=$1; shift
    local s=$(date +%s%N)
    env "$@" KBBQ_TIMING=1 KBBQ_SEED=777 $R/kbbq_amd/kbbq -g $G $D/big.fq.gz > $D/out_$name.gz 2> $D/err_$name.txt || { echo "$name failed"; tail -3 $D/err_$name.txt; exit 1; }
    local e=$(date +%s%N)
    echo "$name wall_ms $(( (e - s) / 1000000 )) $(grep timing $D/err_$name.txt | tr '\n' ' ') out_bytes $(stat -c %s $D/out_$name.gz)" | tee -a $L
}
run device
run device_again
run device_rescan KBBQ_KEEP_TEXT=0
run hostzlib_level1 KBBQ_HOST_DEFLATE=1 KBBQ_BGZF_LEVEL=1
run streaming KBBQ_RESIDENT=0
a=$(gzip -dc $D/out_device.gz | md5sum); b=$(gzip -dc $D/out_hostzlib_level1.gz | md5sum); c=$(gzip -dc $D/out_streaming.gz | md5sum); d=$(gzip -dc $D/out_device_rescan.gz | md5sum)
echo "decompressed md5 device=$a hostzlib=$b streaming=$c device_rescan=$d" | tee -a $L
[ "$a" = "$b" ] && [ "$a" = "$c" ] && [ "$a" = "$d" ] || { echo "OUTPUTS DIFFER" | tee -a $L; exit 1; }
rm -rf $D
