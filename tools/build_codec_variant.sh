#!/bin/bash
# tools/build_codec_variant.sh NAME "-DFLAG=..." -- libkbbq_engine.so with bgzf_device.hip compiled with extra flags, as
# tools/build/libkbbq_NAME.so (the other objects are the tree's own: run `make -C kbbq_amd/csrc` first).  For A/B runs of the
# codec kernels through KBBQ_LIB (tools/inflate_variants.sh, tools/deflate_variants.sh).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/tools/build
cd $R/kbbq_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -Wno-unused-value -Wno-unused-result -ffp-contract=off "$@" \
    -c -o $R/tools/build/bgzf_$name.o bgzf_device.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/build/libkbbq_$name.so engine.o host_model.o $R/tools/build/bgzf_$name.o bgzf_host.o exchange.o -lpthread -ldl
rm -f $R/tools/build/bgzf_$name.o
echo built tools/build/libkbbq_$name.so
