#!/bin/bash
# tools/inflate_variants.sh V1 V2 ... -- tools/inflate_probe.py over the builds tools/build/libkbbq_<V>.so
# (tools/build_codec_variant.sh), each twice in turn in one job: a box differs from the next by more than a variant does
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
    for v in "$@"; do
        echo "variant $v (round $round)"
        KBBQ_LIB=$R/tools/build/libkbbq_$v.so python3 $R/tools/inflate_probe.py 64 16 2>&1 | grep "GB/s\|k_inflate:"
    done
done
