# rocprofv3 kernel table of the training step at the reference's batch (100 x 150): gpurun -- 'TAG=r05 bash tools/train_prof.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/${TAG:-r05}/train_prof
rm -rf $o; mkdir -p $o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -o run -- python3 tools/time_train.py > $o/timing.txt 2> $o/trace.err
grep "train step" $o/timing.txt
f=$(find $o/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
print(", ".join(rows[0][:7]))
for r in rows[1:14]:
    print(", ".join(x[:70] for x in r[:7]))
PY
cp "$f" $o/kernel_stats.csv
