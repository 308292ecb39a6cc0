#!/bin/bash
# tools/r03_e2e_big.sh [TAG] ["RUNS"] -- the command line at 3e10 bases (BASELINE configs[1] at one third): the 3e9-base BGZF FASTQ of
# tools/r03_e2e.sh ten times over (BGZF files concatenate), -g $GENOME; per-phase split, peak host RSS, device budgets.
set -o pipefail
tag=${1:-a}
D=${TMPDIR:-/tmp}/kbbq_e2e_big
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/gpurun_out/r03_e2e_big_$tag.log
mkdir -p $D $R/gpurun_out
: > $L
df -h $D | tail -1 >> $L
python tools/make_fastq.py $D/part.fq 100000000 30 >> $L 2>&1 || exit 1
$R/kbbq_amd/kbbq --io-test bgzf 16 < $D/part.fq > $D/part.fq.gz || exit 1
rm -f $D/part.fq
COPIES=${COPIES:-10}      # COPIES=30: the full BASELINE configs[1] size, 6e8 reads, 9e10 bases (55 GB in; the output only fits a pipe)
GENOME=$(( COPIES * 100000000 ))
for i in $(seq 1 $COPIES); do cat $D/part.fq.gz; done > $D/big.fq.gz
ls -l $D/part.fq.gz $D/big.fq.gz >> $L
echo "files ready"
# first / second: the output through a pipe into wc; file: into a file on the box's disk (what a user does); rescan: pass 4
# reads and inflates the input again (KBBQ_KEEP_TEXT=0: the text of the first scan is not kept in HBM)
for name in ${2:-first second file rescan}; do
    s=$(date +%s%N)
    if [[ $name =~ ^file([0-9]+)$ ]]; then      # fileN: into a file with N writer threads
        KBBQ_WRITE_THREADS=${BASH_REMATCH[1]} KBBQ_TIMING=1 KBBQ_SEED=777 $R/kbbq_amd/kbbq -g $GENOME $D/big.fq.gz 2> $D/err_$name.txt > $D/out.fq.gz || { echo "$name failed"; tail -3 $D/err_$name.txt; exit 1; }
        stat -c %s $D/out.fq.gz > $D/out_$name.bytes; rm -f $D/out.fq.gz
    elif [ $name = file ]; then
        KBBQ_TIMING=1 KBBQ_SEED=777 $R/kbbq_amd/kbbq -g $GENOME $D/big.fq.gz 2> $D/err_$name.txt > $D/out.fq.gz || { echo "$name failed"; tail -3 $D/err_$name.txt; exit 1; }
        stat -c %s $D/out.fq.gz > $D/out_$name.bytes; rm -f $D/out.fq.gz
    elif [ $name = rescan ]; then
        KBBQ_KEEP_TEXT=0 KBBQ_TIMING=1 KBBQ_SEED=777 $R/kbbq_amd/kbbq -g $GENOME $D/big.fq.gz 2> $D/err_$name.txt | wc -c > $D/out_$name.bytes || { echo "$name failed"; tail -3 $D/err_$name.txt; exit 1; }
    else
        KBBQ_TIMING=1 KBBQ_SEED=777 $R/kbbq_amd/kbbq -g $GENOME $D/big.fq.gz 2> $D/err_$name.txt | wc -c > $D/out_$name.bytes || { echo "$name failed"; tail -3 $D/err_$name.txt; exit 1; }
    fi
    e=$(date +%s%N)
    echo "$name wall_ms $(( (e - s) / 1000000 )) $(grep timing $D/err_$name.txt | tr '\n' ' ') out_bytes $(cat $D/out_$name.bytes)" | tee -a $L
    grep -i "resident\|batches" $D/err_$name.txt | head -3 >> $L
done
rm -rf $D
