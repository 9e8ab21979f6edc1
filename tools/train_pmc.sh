# rocprofv3 counters of the training step's kernels at the reference's batch (100 x 150): L2 requests, LDS conflicts, wave cycles
#   gpurun -- 'TAG=r05 bash tools/train_pmc.sh'        (FPC_LIB_PATH selects a variant library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/${TAG:-r05}/train_pmc
rm -rf $o; mkdir -p $o
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $o/$tag -o r -- python3 tools/time_train.py > /dev/null 2> $o/$tag.err || echo "FAILED $c"
done
python3 - <<'PY'
import csv, glob, os, collections, re
o = os.environ.get("TAG", "r05")
base = f"gpurun_out/{o}/train_pmc"
out = open(f"{base}/summary.txt", "w")
def p(*a):
    print(*a); print(*a, file=out)
for d in sorted(glob.glob(f"{base}/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            m = re.search(r"(k_(?:train|forward|grad)\w*)", row.get("Kernel_Name", ""))
            if m:
                acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                p(f"{k:24s} {c:24s} launches {len(v):3d}  mean {sum(v)/len(v):.4g}  max {max(v):.4g}")
            if "SQ_LDS_BANK_CONFLICT" in cs and "SQ_LDS_IDX_ACTIVE" in cs:
                a, b = sum(cs["SQ_LDS_BANK_CONFLICT"]), sum(cs["SQ_LDS_IDX_ACTIVE"])
                p(f"{k:24s} LDS conflict ratio {a / max(b, 1):.3f}")
PY
