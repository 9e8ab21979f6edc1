#!/bin/bash
# build_variants/lib_<name>.so with extra -D flags for lpcnet.hip (kernel experiments; never shipped)
#   tools/build_variant.sh name "-DFPC_DRAW_WAVE=3 ..."
set -e
cd "$(dirname "$0")/../feature-predictor-for-speech-codec_amd/csrc"
make -s
name=$1; shift
out=../../build_variants
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
    -I../../include -Wno-unused-function $* -c lpcnet.hip -o $out/lpcnet_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$name.so api.o predictor.o ceps2lpc.o cb_train.o $out/lpcnet_$name.o
rm -f $out/lpcnet_$name.o
echo built $out/lib_$name.so
