# rocprofv3 evidence for the weights-stationary predictor kernels at 128 x 300: kernel table + LDS / L2 counters
#   gpurun -- 'TAG=r05x bash tools/ws_pmc.sh'        (FPC_LIB_PATH selects a variant library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/${TAG:-r05}/ws_pmc
rm -rf $o; mkdir -p $o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -o run -- python3 tools/ws_time.py --child > $o/timing.txt 2> $o/trace.err
grep -v amdgpu $o/timing.txt
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $o/$tag -o r -- python3 tools/ws_time.py --child > /dev/null 2> $o/$tag.err || echo "FAILED $c"
done
python3 - <<'PY'
import csv, glob, os, collections, re
o = os.environ.get("TAG", "r05")
base = f"gpurun_out/{o}/ws_pmc"
out = open(f"{base}/summary.txt", "w")
def p(*a):
    print(*a); print(*a, file=out)
for f in glob.glob(f"{base}/trace/**/*kernel_stats.csv", recursive=True):
    p("== kernel stats:", f)
    for row in list(csv.reader(open(f)))[:10]:
        p(", ".join(x[:60] for x in row[:8]))
for d in sorted(glob.glob(f"{base}/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            m = re.search(r"(k_(?:encode|forward|hist|decode_feat|train)\w*)", row.get("Kernel_Name", ""))
            if m:
                acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                p(f"{k:30s} {c:24s} launches {len(v):3d}  mean {sum(v)/len(v):.4g}  max {max(v):.4g}")
            if "SQ_LDS_BANK_CONFLICT" in cs and "SQ_LDS_IDX_ACTIVE" in cs:
                a, b = sum(cs["SQ_LDS_BANK_CONFLICT"]), sum(cs["SQ_LDS_IDX_ACTIVE"])
                p(f"{k:30s} LDS conflict ratio {a / max(b, 1):.3f}")
PY
