#!/bin/bash
# build_variants/lib_<name>.so with extra -D flags for predictor.hip (kernel experiments; never shipped)
set -e
cd "$(dirname "$0")/../feature-predictor-for-speech-codec_amd/csrc"
make -s
name=$1; shift
out=../../build_variants
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden \
    -I../../include -Wno-unused-function $* -c predictor.hip -o $out/predictor_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$name.so api.o lpcnet.o ceps2lpc.o cb_train.o kmeans1d.o $out/predictor_$name.o
rm -f $out/predictor_$name.o
echo built $out/lib_$name.so
