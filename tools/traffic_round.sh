#!/bin/bash
# HBM traffic of k_decode on bench.py's default workload (256 x 3 s): separate --pmc passes with --kernel-trace only,
# as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes; writes profiles/${TAG}_traffic.json tied to the kernel's sources
#   gpurun -- 'TAG=r03 bash tools/traffic_round.sh'
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r04}
out=gpurun_out/${TAG}_traffic; mkdir -p $out; rm -rf $out/*
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o runc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2> $out/$c.err || echo "FAILED $c"
done
python3 - <<PY
import csv, json, sys
sys.path.insert(0, ".")
import bench
def mean(counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open("$out/%s/runc_counter_collection.csv" % counter))
         if "k_decode<" in r["Kernel_Name"] and r["Counter_Name"] == counter and int(r["Grid_Size"]) == 256 * 768]  # (the timed step's launches:
    # not k_decode2's, not the 512 / 1 024-workgroup rounds of the many_stream leg)
    return sum(v) / len(v), len(v)
f, nf = mean("FETCH_SIZE")
w, nw = mean("WRITE_SIZE")
rec = {"kernel": "k_decode", "workload": "256 streams x 300 frames (bench.py default)", "kernel_source_sha256": bench.decode_kernel_hash(),
       "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "fetch_correction": 2.0, "launches_averaged": [nf, nw],
       "hbm_bytes_per_launch": int(round((2.0 * f + w) * 1024)),
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (256-workgroup launches of bench.py's default workload). "
               "FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (TCC_EA0_RDREQ tallied at 64 B per 128-B request); "
               "the cold bytes the launch must read are cfa 353.9 MB + cfb 14.7 MB + features 11.1 MB = 379.7 MB."}
json.dump(rec, open("gpurun_out/${TAG}_traffic.json", "w"), indent=1)
print(json.dumps(rec)[:400])
PY
