#!/bin/bash
# tools/e2e.sh TAG GENOME_LEN "CASES" [DIR] -- the command line end to end on the bench's OWN reads as files
# (kbbq --io-test synth-fastq / synth-bam: k_synth -> record text / BAM records -> k_deflate on the device), 30x of
# GENOME_LEN, sampler seed 777 as in bench.py: per-phase split (KBBQ_TIMING=1), the insert counts of the log, the digest of
# the recalibrated qualities taken on the device (KBBQ_QUAL_DIGEST=1) -- to be compared with bench.py's
# result.sampled_inserted / trusted_inserted / recal_qual_sum at the same size -- peak host RSS, output size.
# CASES: fastq bam bam_setoq bamoq_useoq_setoq ; suffix _host = the host parsers (KBBQ_DEVICE_READER=0) ; _file = output into a
# regular file instead of a pipe ; _md5 = also the md5
# of the decompressed output (small sizes only).  Log: gpurun_out/r04_e2e_TAG.log
set -o pipefail
tag=$1; G=${2:-100000000}; cases=${3:-"fastq bam bam_setoq bamoq_useoq_setoq"}
D=${4:-${TMPDIR:-/tmp}/kbbq_e2e_r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/gpurun_out/r04_e2e_$tag.log
mkdir -p $D $R/gpurun_out
: > $L
df -h $D | tail -1 >> $L
have=""
gen() {      # gen NAME io-test-args...
    local name=$1; shift
    [ -f $D/in.$name ] && return 0
    local s=$(date +%s%N)
    $R/kbbq_amd/kbbq --io-test "$@" > $D/in.$name 2>> $L || { echo "generating $name failed"; tail -3 $L; exit 1; }
    local e=$(date +%s%N)
    echo "generated in.$name: $(stat -c %s $D/in.$name) bytes in $(( (e - s) / 1000000 )) ms" | tee -a $L
}
for c in $cases; do
    base=${c%%_host*}; base=${base%%_md5*}; base=${base%%_file*}
    env_extra=""
    [[ $c == *_host* ]] && env_extra="KBBQ_DEVICE_READER=0"
    # (one input at a time: at 9e10 bases each is about 60 GB)
    case $base in
        fastq) rm -f $D/in.bam $D/in.bamoq; gen fq synth-fastq $G 30; in=$D/in.fq; args="-g $G" ;;
        bam) rm -f $D/in.fq $D/in.bamoq; gen bam synth-bam $G 30; in=$D/in.bam; args="" ;;
        bam_setoq) rm -f $D/in.fq $D/in.bamoq; gen bam synth-bam $G 30; in=$D/in.bam; args="--set-oq" ;;
        bamoq_useoq_setoq) rm -f $D/in.bam $D/in.fq; gen bamoq synth-bam $G 30 oq; in=$D/in.bamoq; args="--use-oq --set-oq" ;;
        bamoq_useoq) rm -f $D/in.bam $D/in.fq; gen bamoq synth-bam $G 30 oq; in=$D/in.bamoq; args="--use-oq" ;;
        *) echo "unknown case $c"; exit 1 ;;
    esac
    # E2E_SETTLE=<seconds>: wait before a run -- the driver clears the memory the process before released, and a run
    # that starts behind a 200 GB one finds its own allocations waiting for that
    [ -n "$E2E_SETTLE" ] && sleep $E2E_SETTLE
    s=$(date +%s%N)
    if [[ $c == *_file* ]]; then      # into a regular file beside the input (four pwrite threads by default) instead of a pipe
        env $env_extra KBBQ_TIMING=1 KBBQ_QUAL_DIGEST=1 KBBQ_SEED=777 $R/kbbq_amd/kbbq $args $in 2> $D/err_$c.txt > $D/out_$c.bin || { echo "$c failed"; tail -3 $D/err_$c.txt; exit 1; }
        stat -c %s $D/out_$c.bin > $D/out_$c.bytes; rm -f $D/out_$c.bin
    elif [[ $c == *_md5* ]]; then
        env $env_extra KBBQ_TIMING=1 KBBQ_QUAL_DIGEST=1 KBBQ_SEED=777 $R/kbbq_amd/kbbq $args $in 2> $D/err_$c.txt | tee >(wc -c > $D/out_$c.bytes) | gzip -dc | md5sum > $D/out_$c.md5 || { echo "$c failed"; tail -3 $D/err_$c.txt; exit 1; }
    else
        env $env_extra KBBQ_TIMING=1 KBBQ_QUAL_DIGEST=1 KBBQ_SEED=777 $R/kbbq_amd/kbbq $args $in 2> $D/err_$c.txt | wc -c > $D/out_$c.bytes || { echo "$c failed"; tail -3 $D/err_$c.txt; exit 1; }
    fi
    e=$(date +%s%N)
    echo "== $c ($args) wall_ms $(( (e - s) / 1000000 )) out_bytes $(cat $D/out_$c.bytes) $( [ -f $D/out_$c.md5 ] && cut -c1-32 $D/out_$c.md5 )" | tee -a $L
    grep -E "timing|digest|Sampled|Trusted|trusted kmers|resident|peak" $D/err_$c.txt | sed 's/^/   /' | tee -a $L
done
rm -rf $D
