#!/bin/bash
# SQ counters of k_decode for one library:  tools/pmc_decode.sh <tag> [path/to/lib.so]
# (separate --pmc passes with --kernel-trace only; the 256-stream x 100-frame probe of tools/stamp_probe.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
[ -n "$2" ] && export FPC_LIB_PATH=$GRAFT_REPO_ROOT/$2
out=gpurun_out/pmc_$tag; mkdir -p $out; rm -rf $out/*
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS"; do
  t=$(echo $c | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$t -o r -- python3 tools/stamp_probe.py 256 > /dev/null 2> $out/$t.err || echo "FAILED $c"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/*/r_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = 256 * (100 * 160 - 17)
print("$tag: per sample and workgroup (means over the launches):")
for k in sorted(acc):
    v = sum(acc[k]) / len(acc[k])
    print(f"  {k:24s} {v / n:10.1f}")
PY
