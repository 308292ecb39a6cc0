#!/bin/bash
# tools/sanitize_cli.sh -- the host I/O of the command line (FASTQ reader, block-parallel parser, BGZF, BAM codec) under
# AddressSanitizer + UBSan and under ThreadSanitizer, on the CPU (GPU sanitizers are not available on this pool): builds
# kbbq with g++ -fsanitize=..., runs the --io-test based tests against it (KBBQ_CLI).  Needs the engine library built.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=${TMPDIR:-/tmp}/kbbq_san
mkdir -p $O
cd $R/kbbq_amd/csrc
for mode in address,undefined thread; do
    tag=${mode%%,*}
    g++ -O1 -g -std=c++17 -fsanitize=$mode -fno-omit-frame-pointer -o $O/kbbq_$tag kbbq_cli.cc fastq_io.cc bam_io.cc host_model.cc \
        -I../../include -L.. -lkbbq_engine -lz -lpthread -Wl,-rpath,$R/kbbq_amd -Wl,-rpath,/opt/rocm/lib
done
cd $R
KBBQ_CLI=$O/kbbq_address ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_cli_io_cpu.py tests/test_bam_io_cpu.py -x -q
KBBQ_CLI=$O/kbbq_thread TSAN_OPTIONS=halt_on_error=1 python -m pytest tests/test_cli_io_cpu.py tests/test_bam_io_cpu.py -x -q
