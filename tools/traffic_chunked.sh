#!/bin/bash
# HBM traffic counters of the decode launches with the pass chunked (FPC_LPCNET_CHUNK frames): the sum over a call's launches,
# to set beside tools/traffic_round.sh's one-pass figure.   gpurun -- 'CHUNK=100 bash tools/traffic_chunked.sh'
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CHUNK=${CHUNK:-100}
out=gpurun_out/traffic_chunk$CHUNK; mkdir -p $out; rm -rf $out/*
export FPC_LPCNET_CHUNK=$CHUNK
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o runc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $out/$c.err || echo "FAILED $c"
done
python3 - <<PY
import csv
def tot(counter, pat):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open("$out/%s/runc_counter_collection.csv" % counter))
         if pat in r["Kernel_Name"] and r["Counter_Name"] == counter and int(r["Grid_Size"]) >= 256 * 256]
    return sum(v), len(v)
for pat in ("k_decode", "k_frame_mfma"):
    f, nf = tot("FETCH_SIZE", pat)
    w, nw = tot("WRITE_SIZE", pat)
    print(f"chunk $CHUNK {pat}: {nf} launches over 3 calls; per call FETCH_SIZE x2 {2 * f / 3 / 1024:.1f} MB, WRITE_SIZE {w / 3 / 1024:.1f} MB, sum {(2 * f + w) / 3 / 1024:.1f} MB")
PY
