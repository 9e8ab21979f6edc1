# rocprofv3 evidence for the predictor kernels at 128 x 300 (BASELINE config 5's per-GPU encode share)
#   gpurun -- 'TAG=r02 bash tools/profile_encode.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/${TAG:-r02}/enc_prof
mkdir -p $o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -o run -- python3 tools/time_encode_split.py > $o/timing.txt 2> $o/trace.err
grep -v amdgpu $o/timing.txt
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $o/$tag -o r -- python3 tools/time_encode_split.py > /dev/null 2> $o/$tag.err || echo "FAILED $c"
done
python3 - <<'PY'
import csv, glob, os, collections, re
o = os.environ.get("TAG", "r02")
base = f"gpurun_out/{o}/enc_prof"
out = open(f"{base}/summary.txt", "w")
def p(*a):
    print(*a); print(*a, file=out)
for f in glob.glob(f"{base}/trace/**/*kernel_stats.csv", recursive=True):
    p("== kernel stats:", f)
    for row in list(csv.reader(open(f)))[:8]:
        p(", ".join(row[:8]))
for d in sorted(glob.glob(f"{base}/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            m = re.search(r"(k_(?:encode|forward|hist|decode_feat)\w*)", row.get("Kernel_Name", ""))
            if m:
                acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                p(f"{k:42s} {c:24s} launches {len(v):3d}  mean {sum(v)/len(v):.4g}  max {max(v):.4g}")
PY
