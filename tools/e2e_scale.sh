#!/bin/bash
# tools/e2e_scale.sh [GENOME_LEN] -- the command line end to end at a scale past 2^32 bytes of input, checked
# against the engine's own resident-batch path: the sum of the recalibrated qualities in the FASTQ it writes
# must equal bench.py's digest for the same synthetic reads.  Logs under gpurun_out/.
set -o pipefail
G=${1:-100000000}
D=${TMPDIR:-/tmp}/kbbq_e2e
mkdir -p $D gpurun_out
python tools/make_fastq.py $D/big.fq $G 30 > gpurun_out/e2e_scale.log 2>&1 || exit 1
ls -l $D/big.fq >> gpurun_out/e2e_scale.log
s=$(date +%s%N)
KBBQ_SEED=777 kbbq_amd/kbbq -g $G $D/big.fq > $D/out.fq.gz 2>> gpurun_out/e2e_scale.log || { echo "kbbq failed"; exit 1; }
e=$(date +%s%N)
echo "cli_ms $(( (e - s) / 1000000 ))" | tee -a gpurun_out/e2e_scale.log
python - $D/out.fq.gz <<'PY' | tee -a gpurun_out/e2e_scale.log
import subprocess, sys
import numpy as np
W = 319                                   # @r0000000000/1 \n 150 \n + \n 150 \n  (tools/make_fastq.py)
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
print("cli_records %d cli_qual_sum %d" % (recs, total))
PY
python bench.py --genome-len $G --no-cpu-baseline --steps 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench_qual_sum', d['result']['recal_qual_sum'], 'reads', d['config']['reads'])" | tee -a gpurun_out/e2e_scale.log
rm -rf $D
