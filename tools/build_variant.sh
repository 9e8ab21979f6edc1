#!/bin/bash
# build_variants/lib_<name>.so = the library with extra compiler flags (diagnostic / tuning builds; FPC_LIB_PATH selects one)
#   bash tools/build_variant.sh ws_prof -DFPC_WS_PROF
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/feature-predictor-for-speech-codec_amd/csrc
out=$root/build_variants/obj_$name
mkdir -p $out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$root/include -I$src -Wall -Wno-unused-function"
for f in api lpcnet predictor ceps2lpc cb_train kmeans1d; do
  # (every object, every time: a stale object from an older tree once shipped a variant without a new export)
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $src/$f.hip -o $out/$f.o 2>&1 | grep -v 'argument unused' || true
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build_variants/lib_$name.so $out/*.o
echo built build_variants/lib_$name.so
