#!/bin/bash
# one iteration of the k_decode2 tuning loop on the GPU box: parity (small), timing at B = 512, phase stamps
# usage (on the box): bash tools/pair_iter.sh <tag>
tag=${1:-x}
mkdir -p gpurun_out
{
  timeout -k 10 300 python tools/pair_probe.py --T 100 --B 512 2>&1 | grep -v amdgpu.ids
  echo "== stamps (pairing 1, B=512)"
  FPC_LPCNET_PAIRING=1 FPC_DECODE_STAMPS=1 timeout -k 10 200 python tools/stamp_probe.py 512 2>&1 | grep -E "phase lengths|wave  [014] |wave  8|wave 10|decode ms"
} > gpurun_out/pair_iter_$tag.log 2>&1
cat gpurun_out/pair_iter_$tag.log
