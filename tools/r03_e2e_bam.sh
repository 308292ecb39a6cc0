#!/bin/bash
# tools/r03_e2e_bam.sh [GENOME_LEN] -- the command line end to end on unaligned BAM (BASELINE configs[3]'s shape at the size
# given: 30x of GENOME_LEN, RG:Z tags, half the records reverse-flagged): per-phase split (KBBQ_TIMING=1) with and without
# --set-oq, and the sum of the recalibrated qualities in the BAM it writes against bench.py's digest for the same reads.
# Also the FASTQ digest check of tools/e2e_scale.sh with the block-parallel parser.  Log: gpurun_out/r03_e2e_bam.log
set -o pipefail
G=${1:-100000000}
D=${TMPDIR:-/tmp}/kbbq_e2e
R=${GRAFT_REPO_ROOT:-$(pwd)}
L=$R/gpurun_out/r03_e2e_bam.log
mkdir -p $D $R/gpurun_out
: > $L
python tools/make_bam.py $D/big.bam $G 30 >> $L 2>&1 || exit 1
ls -l $D/big.bam >> $L
run() {
    local name=$1; shift
    local s=$(date +%s%N)
    KBBQ_TIMING=1 KBBQ_SEED=777 $R/kbbq_amd/kbbq "$@" $D/big.bam > $D/out_$name.bam 2> $D/err_$name.txt || { echo "$name failed"; tail -3 $D/err_$name.txt; exit 1; }
    local e=$(date +%s%N)
    echo "bam $name wall_ms $(( (e - s) / 1000000 )) $(grep timing $D/err_$name.txt) out_bytes $(stat -c %s $D/out_$name.bam)" | tee -a $L
}
run plain
run set_oq --set-oq
python - $D/out_plain.bam <<'PY' | tee -a $L
import struct, subprocess, sys
import numpy as np
p = subprocess.Popen(["gzip", "-dc", sys.argv[1]], stdout=subprocess.PIPE)
rd = p.stdout.read
head = rd(8); assert head[:4] == b"BAM\1"
rd(struct.unpack("<I", head[4:])[0])
for _ in range(struct.unpack("<I", rd(4))[0]):
    rd(struct.unpack("<I", rd(4))[0] + 4)
W = 4 + 32 + 12 + 75 + 150 + 9          # tools/make_bam.py: fixed-size records
total = recs = 0
while True:
    buf = rd(W * 500000)
    if not buf:
        break
    a = np.frombuffer(buf, dtype=np.uint8).reshape(-1, W)
    assert (a[:, 0:4] == np.frombuffer(struct.pack("<I", W - 4), dtype=np.uint8)).all()
    total += int(a[:, 123:273].astype(np.int64).sum())
    recs += a.shape[0]
print("bam_records %d bam_qual_sum %d" % (recs, total))
PY
rm -f $D/big.bam $D/out_*.bam
python tools/make_fastq.py $D/big.fq $G 30 >> $L 2>&1 || exit 1
s=$(date +%s%N)
$R/kbbq_amd/kbbq --io-test bgzf 16 < $D/big.fq > $D/big.fq.gz; rm -f $D/big.fq; KBBQ_TIMING=1 KBBQ_SEED=777 $R/kbbq_amd/kbbq -g $G $D/big.fq.gz > $D/out.fq.gz 2>> $L || { echo "kbbq failed"; exit 1; }
e=$(date +%s%N)
echo "fastq cli_ms $(( (e - s) / 1000000 ))" | tee -a $L
python - $D/out.fq.gz <<'PY' | tee -a $L
import subprocess, sys
import numpy as np
W = 319
p = subprocess.Popen(["gzip", "-dc", sys.argv[1]], stdout=subprocess.PIPE)
total = recs = 0
while True:
    buf = p.stdout.read(W * 1000000)
    if not buf:
        break
    a = np.frombuffer(buf, dtype=np.uint8).reshape(-1, W)
    assert (a[:, 0] == ord("@")).all() and (a[:, W - 1] == 10).all()
    total += int(a[:, 168:318].astype(np.int64).sum()) - 33 * 150 * a.shape[0]
    recs += a.shape[0]
print("fastq_records %d fastq_qual_sum %d" % (recs, total))
PY
python bench.py --genome-len $G --no-cpu-baseline --no-pcie --steps 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench_qual_sum', d['result']['recal_qual_sum'], 'reads', d['config']['reads'])" | tee -a $L
rm -rf $D
