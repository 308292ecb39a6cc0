#!/bin/bash
# tools/deflate_variants.sh -- tools/deflate_probe.py over the builds of bgzf_device.hip in tools/build/ (libkbbq_w<WAYS>b<BITS>.so:
# -DKBBQ_DFL_WAYS / -DKBBQ_DFL_HASH_BITS), each twice in turn, and zlib level 6 on the same text: size and rate per variant
R=${GRAFT_REPO_ROOT:-$(pwd)}
KBBQ_PROBE_ZLIB=1 KBBQ_LIB=$R/tools/build/libkbbq_w1b12.so python3 $R/tools/deflate_probe.py 64 8 | head -2
for round in 1 2; do
    for v in w1b12 w2b10 w2b11 w2b12; do
        echo "variant $v (round $round)"
        KBBQ_LIB=$R/tools/build/libkbbq_$v.so python3 $R/tools/deflate_probe.py 64 8 | tail -2
    done
done
